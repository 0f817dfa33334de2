// Debug visitor every unit's dump() writes into (reference interface: include/lsp-plug.in/dsp-units/iface/IStateDumper.h:36-222).
// Every method is a no-op by default; a host overrides what it wants to see.  The virtual methods are declared in the
// reference's order -- objects, arrays, unnamed scalars, named scalars, unnamed vectors, named vectors, each over the same
// list of arithmetic types -- so that a dumper written against the reference's header sees the same slots.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_IFACE_ISTATEDUMPER_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_IFACE_ISTATEDUMPER_H_

#include <lsp-plug.in/dsp-units/version.h>
#include <cstddef>
#include <cstdint>

// the arithmetic types a dumper distinguishes, in slot order
#define MI_DUMPER_TYPES(X) \
    X(bool) X(unsigned char) X(signed char) X(unsigned short) X(signed short) X(unsigned int) X(signed int) \
    X(unsigned long) X(signed long) X(unsigned long long) X(signed long long) X(float) X(double)

namespace lsp
{
    namespace dspu
    {
        class LSP_DSP_UNITS_PUBLIC IStateDumper
        {
            public:
                explicit IStateDumper() {}
                virtual ~IStateDumper() {}
                IStateDumper(const IStateDumper &) = delete;
                IStateDumper(IStateDumper &&) = delete;
                IStateDumper &operator = (const IStateDumper &) = delete;
                IStateDumper &operator = (IStateDumper &&) = delete;

            public:
                // structure
                virtual void begin_object(const char *, const void *, size_t) {}
                virtual void begin_object(const void *, size_t) {}
                virtual void end_object() {}
                virtual void begin_array(const char *, const void *, size_t) {}
                virtual void begin_array(const void *, size_t) {}
                virtual void end_array() {}

                // one value, without and with a name
                virtual void write(const void *) {}
                virtual void write(const char *) {}
                #define MI_DUMPER_SLOT(T) virtual void write(T) {}
                MI_DUMPER_TYPES(MI_DUMPER_SLOT)
                #undef MI_DUMPER_SLOT
                virtual void write(const char *, const void *) {}
                virtual void write(const char *, const char *) {}
                #define MI_DUMPER_SLOT(T) virtual void write(const char *, T) {}
                MI_DUMPER_TYPES(MI_DUMPER_SLOT)
                #undef MI_DUMPER_SLOT

                // `count` values, without and with a name
                virtual void writev(const void * const *, size_t) {}
                #define MI_DUMPER_SLOT(T) virtual void writev(const T *, size_t) {}
                MI_DUMPER_TYPES(MI_DUMPER_SLOT)
                #undef MI_DUMPER_SLOT
                virtual void writev(const char *, const void * const *, size_t) {}
                #define MI_DUMPER_SLOT(T) virtual void writev(const char *, const T *, size_t) {}
                MI_DUMPER_TYPES(MI_DUMPER_SLOT)
                #undef MI_DUMPER_SLOT

            public:
                // tables of pointers to anything go out as tables of addresses
                template <class T> void writev(const T * const *tab, size_t count)
                    { writev(reinterpret_cast<const void * const *>(tab), count); }
                template <class T> void writev(const char *name, const T * const *tab, size_t count)
                    { writev(name, reinterpret_cast<const void * const *>(tab), count); }

                // a unit that has a dump() of its own: an object around its dump, or its (null) address
                template <class T> void write_object(const T *unit)
                {
                    if (unit == nullptr) { write(static_cast<const void *>(unit)); return; }
                    begin_object(unit, sizeof(T));
                    unit->dump(this);
                    end_object();
                }
                template <class T> void write_object(const char *name, const T *unit)
                {
                    if (unit == nullptr) { write(name, static_cast<const void *>(unit)); return; }
                    begin_object(name, unit, sizeof(T));
                    unit->dump(this);
                    end_object();
                }

                // `count` such units side by side ...
                template <class T> void write_object_array(const T *units, size_t count)
                {
                    if (units == nullptr) { write(static_cast<const void *>(units)); return; }
                    begin_array(units, count);
                    for (size_t i = 0; i < count; ++i)
                        write_object(units + i);
                    end_array();
                }
                template <class T> void write_object_array(const char *name, const T *units, size_t count)
                {
                    if (units == nullptr) { write(name, static_cast<const void *>(units)); return; }
                    begin_array(name, units, count);
                    for (size_t i = 0; i < count; ++i)
                        write_object(units + i);
                    end_array();
                }
                // ... or behind a table of pointers
                template <class T> void write_object_array(const T * const *tab, size_t count)
                {
                    if (tab == nullptr) { write(static_cast<const void *>(tab)); return; }
                    begin_array(tab, count);
                    for (size_t i = 0; i < count; ++i)
                        write_object(tab[i]);
                    end_array();
                }
                template <class T> void write_object_array(const char *name, const T * const *tab, size_t count)
                {
                    if (tab == nullptr) { write(name, static_cast<const void *>(tab)); return; }
                    begin_array(name, tab, count);
                    for (size_t i = 0; i < count; ++i)
                        write_object(tab[i]);
                    end_array();
                }
        };
    }
}

#endif
