// Debug visitor every unit's dump() writes into; all methods are no-ops by default.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_IFACE_ISTATEDUMPER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_IFACE_ISTATEDUMPER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <cstddef>
#include <cstdint>

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC IStateDumper
        {
            public:
                virtual ~IStateDumper() {}
                virtual void begin_object(const char *, const void *, size_t) {}
                virtual void end_object() {}
                virtual void write(const char *, const void *) {}
                virtual void write(const char *, size_t) {}
                virtual void write(const char *, float) {}
                virtual void write(const char *, bool) {}
        };
    }
}

#endif
