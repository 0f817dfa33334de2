// Unit conversions used by callers of the hot path (decibels <-> gain, time <-> samples).
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_UNITS_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_UNITS_H_

#include <cmath>

namespace lsp
{
    namespace dspu
    {
        inline float db_to_gain(float db)       { return expf(db * float(M_LN10) * 0.05f); }
        inline float gain_to_db(float gain)     { return (20.0f / float(M_LN10)) * logf(gain); }
        inline float seconds_to_samples(float sr, float time)   { return time * sr; }
        inline float samples_to_seconds(float sr, float samples) { return samples / sr; }
        inline float millis_to_samples(float sr, float time)    { return (time * 0.001f) * sr; }
    }
}

#endif
