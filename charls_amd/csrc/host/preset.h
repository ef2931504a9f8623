// preset.h -- JPEG-LS preset coding parameters: defaults (ISO 14495-1 C.2.4.1.1.1) and validation.
// Behaviour of reference src/jpegls_preset_coding_parameters.hpp:24-130.
#pragma once
#include <algorithm>

#include "common.h"

namespace jls {

inline int32_t clamp_threshold(int32_t i, int32_t j, int32_t maxval)
{
    return (i > maxval || i < j) ? j : i;
}

inline charls_jpegls_pc_parameters default_pc(int32_t maxval, int32_t near)
{
    int32_t t1, t2, t3;
    if (maxval >= 128)
    {
        const int32_t f = (std::min(maxval, 4095) + 128) / 256;
        t1 = clamp_threshold(f + 2 + 3 * near, near + 1, maxval);
        t2 = clamp_threshold(f * 4 + 3 + 5 * near, t1, maxval);
        t3 = clamp_threshold(f * 17 + 4 + 7 * near, t2, maxval);
    }
    else
    {
        const int32_t f = 256 / (maxval + 1);
        t1 = clamp_threshold(std::max(2, 3 / f + 3 * near), near + 1, maxval);
        t2 = clamp_threshold(std::max(3, 7 / f + 5 * near), t1, maxval);
        t3 = clamp_threshold(std::max(4, 21 / f + 7 * near), t2, maxval);
    }
    return {maxval, t1, t2, t3, 64};
}

inline bool pc_is_default(const charls_jpegls_pc_parameters& p, const charls_jpegls_pc_parameters& d)
{
    if (!p.maximum_sample_value && !p.threshold1 && !p.threshold2 && !p.threshold3 && !p.reset_value)
        return true;
    return p.maximum_sample_value == d.maximum_sample_value && p.threshold1 == d.threshold1 &&
           p.threshold2 == d.threshold2 && p.threshold3 == d.threshold3 && p.reset_value == d.reset_value;
}

// Zero members mean "default"; `out` receives the values the scan codec must use.
inline bool pc_validate(const charls_jpegls_pc_parameters& p, int32_t bit_maxval, int32_t near,
                        charls_jpegls_pc_parameters* out)
{
    if (p.maximum_sample_value != 0 && (p.maximum_sample_value < 1 || p.maximum_sample_value > bit_maxval))
        return false;
    const int32_t maxval = p.maximum_sample_value != 0 ? p.maximum_sample_value : bit_maxval;
    if (p.threshold1 != 0 && (p.threshold1 < near + 1 || p.threshold1 > maxval))
        return false;
    const charls_jpegls_pc_parameters d = default_pc(maxval, near);
    const int32_t t1 = p.threshold1 != 0 ? p.threshold1 : d.threshold1;
    if (p.threshold2 != 0 && (p.threshold2 < t1 || p.threshold2 > maxval))
        return false;
    const int32_t t2 = p.threshold2 != 0 ? p.threshold2 : d.threshold2;
    if (p.threshold3 != 0 && (p.threshold3 < t2 || p.threshold3 > maxval))
        return false;
    if (p.reset_value != 0 && (p.reset_value < 3 || p.reset_value > std::max(255, maxval)))
        return false;
    if (out)
        *out = {maxval, t1, t2, p.threshold3 != 0 ? p.threshold3 : d.threshold3,
                p.reset_value != 0 ? p.reset_value : d.reset_value};
    return true;
}

// src/color_transform.hpp:11-16
inline bool color_transformation_possible(const charls_frame_info& f, int32_t near, int32_t ilv)
{
    return f.component_count == 3 && (f.bits_per_sample == 8 || f.bits_per_sample == 16) && near == 0 && ilv != 0;
}

} // namespace jls
