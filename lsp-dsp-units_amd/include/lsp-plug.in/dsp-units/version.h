#ifndef MI_LSP_PLUG_IN_DSP_UNITS_VERSION_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_VERSION_H_

// API level mirrored from lsp-dsp-units
#define LSP_DSP_UNITS_MAJOR         1
#define LSP_DSP_UNITS_MINOR         0
#define LSP_DSP_UNITS_MICRO         36

#if defined(__GNUC__)
    #define LSP_DSP_UNITS_PUBLIC    __attribute__((visibility("default")))
#else
    #define LSP_DSP_UNITS_PUBLIC
#endif

namespace lsp
{
    namespace dspu
    {
        // The class API has no error channel (process() is void, as in the reference).  The last non-zero MI_* status
        // of a device call made on behalf of an lsp::dspu object of the calling thread is kept here (0 = none since
        // clear_last_status()); mi_dspu_last_error() of mi_dspu.h holds its text.
        LSP_DSP_UNITS_PUBLIC int    last_status();
        LSP_DSP_UNITS_PUBLIC void   clear_last_status();
    }
}

#endif
