// misc_api.cpp -- the five remaining reference exports (src/jpegls_error.cpp, src/version.cpp,
// src/validate_spiff_header.cpp) and the engine-control extensions of charls_amd.h part 2.
#include <string>
#include <system_error>

#include "../device/knobs.h"
#include "../device/runtime.h"
#include "common.h"
#include "scan_engine.h"

using namespace jls;

namespace {

const char* message_of(int32_t e) noexcept
{
    switch (e)
    {
    case 0: return "Success";
    case 1: return "Not enough memory (host or GPU) to complete the operation";
    case 2: return "A callback function returned a non-zero value";
    case 3: return "The destination buffer is too small to hold all the output";
    case 4: return "The source buffer ends before the JPEG-LS stream is complete";
    case 5: return "The entropy-coded data is not valid JPEG-LS data";
    case 6: return "The stream uses a JPEG encoding other than JPEG-LS";
    case 7: return "The stream uses a parameter value this implementation does not support";
    case 8: return "The colour transformation of the stream is not supported";
    case 9: return "The stream uses a JPEG-LS extended (ISO/IEC 14495-2) preset parameter type";
    case 10: return "A marker was expected but the 0xFF start byte is missing";
    case 11: return "The stream does not start with a start-of-image (SOI) marker";
    case 12: return "The SPIFF header is not valid";
    case 13: return "The stream contains an unknown JPEG marker";
    case 14: return "A start-of-scan (SOS) marker appears before the frame header";
    case 15: return "A marker segment has an invalid size";
    case 16: return "The stream contains more than one start-of-image (SOI) marker";
    case 17: return "The stream contains more than one start-of-frame (SOF) marker";
    case 18: return "The frame header lists the same component identifier twice";
    case 19: return "An end-of-image (EOI) marker appears before the image data";
    case 20: return "The JPEG-LS preset parameters segment has an invalid type";
    case 21: return "The SPIFF directory is not terminated by an end-of-directory entry";
    case 22: return "A restart marker appears outside entropy-coded data";
    case 23: return "The expected restart marker is missing";
    case 24: return "The end-of-image (EOI) marker is missing";
    case 25: return "A define-number-of-lines (DNL) marker appears where it is not allowed";
    case 26: return "The frame height is zero and no define-number-of-lines (DNL) marker follows the first scan";
    case 27: return "A scan header refers to a component identifier that is not in the frame header";
    case 28: return "A SPIFF header cannot be combined with an abbreviated format for table specification data";
    case 29: return "The stream defines an invalid width";
    case 30: return "The stream defines an invalid height";
    case 31: return "The stream defines an invalid number of bits per sample";
    case 32: return "The stream defines an invalid component count";
    case 33: return "The stream defines an invalid interleave mode";
    case 34: return "The stream defines an invalid NEAR value";
    case 35: return "The stream defines invalid JPEG-LS preset coding parameters";
    case 36: return "The stream defines an invalid colour transformation";
    case 37: return "The stream defines an invalid mapping table identifier";
    case 38: return "The stream continues a mapping table that was not started or has another entry size";
    case 100: return "The function cannot be called in the current state of the object";
    case 101: return "An argument is invalid";
    case 102: return "The width argument is outside [1, 100000]";
    case 103: return "The height argument is outside [1, 100000]";
    case 104: return "The bits-per-sample argument is outside [2, 16]";
    case 105: return "The component-count argument is outside [1, 255]";
    case 106: return "The interleave-mode argument is invalid for this frame";
    case 107: return "The NEAR argument is outside [0, min(255, MAXVAL/2)]";
    case 108: return "The JPEG-LS preset coding parameters argument is invalid";
    case 109: return "The colour transformation argument is invalid for this frame";
    case 110: return "The size argument is invalid";
    case 111: return "The stride argument is smaller than one row of pixels";
    case 112: return "The encoding-options argument has unknown bits set";
    case 200: return "No usable gfx950 GPU: charls_amd has no CPU fallback";
    case 201: return "The GPU runtime reported a failure";
    default: return "Unknown charls_jpegls_errc value";
    }
}

class jpegls_category_t final : public std::error_category
{
public:
    const char* name() const noexcept override { return "charls::jpegls"; }
    std::string message(int v) const override { return message_of(v); }
};

} // namespace

extern "C" {

const char* charls_get_error_message(charls_jpegls_errc error_value)
{
    return message_of(error_value);
}

const void* charls_get_jpegls_category(void)
{
    static const jpegls_category_t instance;
    return static_cast<const std::error_category*>(&instance);
}

const char* charls_get_version_string(void)
{
    return "3.0.0"; // ABI level of the reference this engine mirrors (include/charls/version.h:12-14)
}

void charls_get_version_number(int32_t* major, int32_t* minor, int32_t* patch)
{
    if (major)
        *major = 3;
    if (minor)
        *minor = 0;
    if (patch)
        *patch = 0;
}

charls_jpegls_errc charls_validate_spiff_header(const charls_spiff_header* h, const charls_frame_info* f)
{ // reference src/validate_spiff_header.cpp:12-92
    if (!h || !f)
        return CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT;
    bool ok = h->compression_type == 6 && h->profile_id == 0 && h->resolution_units >= 0 && h->resolution_units <= 2 &&
              h->horizontal_resolution != 0 && h->vertical_resolution != 0 && h->component_count == f->component_count &&
              h->bits_per_sample == f->bits_per_sample && h->height == f->height && h->width == f->width;
    if (ok)
    {
        switch (h->color_space)
        {
        case 2: // none
            break;
        case 8: // grayscale
            ok = h->component_count == 1;
            break;
        case 1: case 3: case 4: case 10: case 11: case 9: case 14: // YCbCr x3, RGB, CMY, PhotoYCC, CIELab
            ok = h->component_count == 3;
            break;
        case 12: case 13: // CMYK, YCCK
            ok = h->component_count == 4;
            break;
        default: // bi-level and unknown values
            ok = false;
        }
    }
    return ok ? CHARLS_JPEGLS_ERRC_SUCCESS : CHARLS_JPEGLS_ERRC_INVALID_SPIFF_HEADER;
}

charls_jpegls_errc charls_amd_device_status(void)
{
    return dev::device_status();
}

charls_jpegls_errc charls_amd_set_encode_engine(int32_t engine)
{
    if (engine < 0 || engine > 2)
        return CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT;
    dev::set_encode_engine(static_cast<dev::EncodeEngine>(engine));
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}

charls_jpegls_errc charls_amd_set_workspace_limit(uint64_t bytes)
{
    dev::set_workspace_limit(bytes);
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}

charls_jpegls_errc charls_amd_release_work_areas(void)
{
    dev::release_work_areas();
    release_idle_engine_resources(); // the device buffers, streams and staging areas that handles share (scan_engine.h)
    return CHARLS_JPEGLS_ERRC_SUCCESS;
}

uint64_t charls_amd_work_area_bytes(void)
{
    return dev::work_area_bytes();
}

int32_t charls_amd_speculation_counters(uint64_t* out, int32_t capacity)
{
    uint64_t v[dev::tile_counter_count];
    dev::speculation_counters(v);
    int32_t n = 0;
    for (; out != nullptr && n < capacity && n < dev::tile_counter_count; ++n)
        out[n] = v[n];
    return n;
}

int32_t charls_amd_engine_counters(uint64_t* out, int32_t capacity)
{
    uint64_t stats[5], v[10];
    coalescer_stats(stats);
    for (int i = 0; i < 4; ++i)
        v[i] = stats[i];
    v[4] = dev::pipeline_fallback_scans();
    v[5] = stats[4];
    v[6] = idle_engine_resource_bytes();
    v[7] = dev::deferred_free_bytes();
    v[8] = idle_releases();
    v[9] = dev::exact_retry_scans();
    int32_t n = 0;
    for (; out != nullptr && n < capacity && n < 10; ++n)
        out[n] = v[n];
    return n;
}

charls_jpegls_errc charls_amd_debug_set_knob(const char* name, int64_t value)
{
    return knobs::set(name, value == INT64_MIN ? knobs::kUnset : static_cast<long long>(value)) ? CHARLS_JPEGLS_ERRC_SUCCESS
                                                                                                 : CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT;
}

int32_t charls_amd_last_timings(double* out, int32_t capacity)
{
    const dev::Timings& t = dev::last_timings();
    const int32_t n = t.count < capacity ? t.count : capacity;
    for (int32_t i = 0; i < n; ++i)
        out[i] = t.values[i];
    return n;
}

} // extern "C"
