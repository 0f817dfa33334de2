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

#endif
