// lsp::dspu::bs -- broadcast-related constants of the reference (misc/broadcast.h) used by the loudness meter.
#ifndef MI_LSP_PLUG_IN_DSP_UNITS_MISC_BROADCAST_H_
#define MI_LSP_PLUG_IN_DSP_UNITS_MISC_BROADCAST_H_

#include <lsp-plug.in/dsp-units/version.h>

namespace lsp
{
    namespace dspu
    {
        namespace bs
        {
            enum weighting_t
            {
                WEIGHT_NONE, WEIGHT_A, WEIGHT_B, WEIGHT_C, WEIGHT_D, WEIGHT_K
            };

            // ITU-R BS.2051-3 channel designations, same order (and values) as the reference
            enum channel_t
            {
                CHANNEL_NONE, CHANNEL_CENTER, CHANNEL_LEFT_SCREEN, CHANNEL_RIGHT_SCREEN, CHANNEL_LEFT, CHANNEL_RIGHT,
                CHANNEL_FRONT_LEFT, CHANNEL_FRONT_RIGHT, CHANNEL_LEFT_SIDE, CHANNEL_RIGHT_SIDE, CHANNEL_LEFT_SURROUND,
                CHANNEL_RIGHT_SURROUND, CHANNEL_LEFT_BACK, CHANNEL_RIGHT_BACK, CHANNEL_BACK_CENTER,
                CHANNEL_TOP_FRONT_CENTER, CHANNEL_LEFT_TOP_FRONT, CHANNEL_RIGHT_TOP_FRONT, CHANNEL_LEFT_HEIGHT,
                CHANNEL_RIGHT_HEIGHT, CHANNEL_TOP_SIDE_LEFT, CHANNEL_TOP_SIDE_RIGHT, CHANNEL_LEFT_TOP_REAR,
                CHANNEL_RIGHT_TOP_REAR, CHANNEL_TOP_BACK_LEFT, CHANNEL_TOP_BACK_RIGHT, CHANNEL_TOP_BACK_CENTER,
                CHANNEL_CENTER_HEIGHT, CHANNEL_TOP_CENTER, CHANNEL_CENTER_BOTTOM_FRONT, CHANNEL_BOTTOM_FRONT_LEFT,
                CHANNEL_BOTTOM_FRONT_RIGHT, CHANNEL_LFE1, CHANNEL_LFE2
            };

            constexpr float DBFS_TO_LUFS_SHIFT_DB   = -0.691f;
            constexpr float LUFS_TO_DBFS_SHIFT_DB   =  0.691f;
            constexpr float LUFS_TO_LU_SHIFT_DB     = 23.0f;
            constexpr float LO_TO_LUFS_SHIFT_DB     = -23.0f;
            constexpr float DB_TO_LU_SHIFT_DB       = 22.309f;
            constexpr float LU_TO_DB_SHIFT          = -22.309f;
            constexpr float DBFS_TO_LUFS_SHIFT_GAIN = 0.923527857225f;      // 10^(-0.691 / 20)
            constexpr float LUFS_TO_DBFS_SHIFT_GAIN = 1.08280437041f;
            constexpr float LUFS_TO_LU_SHIFT_GAIN   = 14.1253754462f;       // 10^(23 / 20)
            constexpr float LO_TO_LUFS_SHIFT_GAIN   = 0.0707945784385f;
            constexpr float DB_TO_LU_SHIFT_GAIN     = 13.0451777184f;
            constexpr float LU_TO_DB_SHIFT_GAIN     = 0.0766566789345f;
            constexpr float LUFS_MEASURE_PERIOD_MS  = 400.0f;
            constexpr float LUFS_MOMENTARY_PERIOD   = 400.0f;
            constexpr float LUFS_SHORT_TERM_PERIOD  = 3000.0f;

            LSP_DSP_UNITS_PUBLIC
            float channel_weighting(channel_t designation);
        }
    }
}

#endif
