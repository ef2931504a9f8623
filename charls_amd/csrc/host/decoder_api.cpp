// decoder_api.cpp -- `struct charls_jpegls_decoder` and its 20 extern "C" entry points.
//
// Same state machine, checks and error codes as the reference facade (src/charls_jpegls_decoder.cpp:21-533); the scan
// itself (`make_scan_codec<scan_decoder>()->decode_scan`, :186-189) runs on the GPU through ScanEngine.
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "scan_engine.h"
#include "stream_reader.h"

using namespace jls;

struct charls_jpegls_decoder
{
    enum class State
    {
        initial,
        source_set,
        spiff_header_read,
        spiff_header_not_found,
        header_read,
        completed
    };

    void check_header_read() const { check_operation(state >= State::header_read); }
    void check_completed() const { check_operation(state == State::completed); }

    size_t minimum_stride() const noexcept // reference :238-244
    {
        const size_t planes = reader.scan_interleave_mode() == 0 ? 1u : static_cast<size_t>(reader.scan_component_count());
        return planes * reader.frame_info().width * bytes_per_sample(reader.frame_info().bits_per_sample);
    }

    size_t destination_size(size_t stride) const // reference :92-121
    {
        check_header_read();
        const charls_frame_info& f = reader.frame_info();
        if (stride == 0)
            return checked_mul(checked_mul(checked_mul(static_cast<size_t>(f.component_count), f.height), f.width),
                               bytes_per_sample(f.bits_per_sample));
        check_argument(reader.component_count() > 0);
        if (reader.interleave_mode(0) == 0)
        {
            const size_t min_stride = static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
            check_argument(stride >= min_stride, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
            return checked_mul(checked_mul(stride, static_cast<size_t>(f.component_count)), f.height) - (stride - min_stride);
        }
        const size_t min_stride =
            static_cast<size_t>(f.width) * static_cast<size_t>(f.component_count) * bytes_per_sample(f.bits_per_sample);
        check_argument(stride >= min_stride, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
        return checked_mul(stride, f.height) - (stride - min_stride);
    }

    // The component scans of a planar frame, decoded by ONE launch (the reference decodes them in a loop, :177-201; they
    // share nothing, and where each one starts can be found without decoding: FF followed by a byte >= 0x80 cannot occur
    // inside entropy-coded data, src/scan_decoder.hpp:272-284).  A copy of the reader -- without the caller's comment /
    // application-data handlers -- walks from scan to scan over the markers in between; the scans must be single-component
    // scans with the same coding parameters.  Returns false, with nothing changed, whenever anything is out of the ordinary
    // (a segment that does not parse, different parameters, a scan that does not end where the next marker is, any decoding
    // error): the scan-by-scan path then runs from the start and raises what the reference would.
    // `uploaded` is set once the source bytes are on the device (the scan-by-scan path does not send them again).
    bool decode_planes_together(uint8_t* dst, size_t dst_left, size_t stride_arg, const uint8_t* base, bool& uploaded)
    {
        const size_t count = reader.component_count();
        if (count < 2 || reader.scan_interleave_mode() != 0 || reader.scan_component_count() != 1)
            return false;
        const charls_frame_info f = reader.frame_info();
        const size_t min_stride = static_cast<size_t>(f.width) * bytes_per_sample(f.bits_per_sample);
        const size_t stride = stride_arg == 0 ? min_stride : stride_arg;
        if (stride < min_stride)
            return false;
        size_t needed = 0;
        try
        {
            needed = checked_mul(checked_mul(stride, f.height), count) - (stride - min_stride);
        }
        catch (const error&)
        {
            return false;
        }
        if (dst_left < needed)
            return false;
        std::vector<size_t> offsets, marker_at; // where scan c starts; where the marker that ends it was found
        StreamReader after; // the probe behind the last scan's header
        std::vector<ScanSpec> specs;
        try
        {
            StreamReader probe = reader;
            probe.at_comment(nullptr, nullptr);
            probe.at_application_data(nullptr, nullptr);
            for (size_t c = 0; c < count; ++c)
            {
                if (probe.scan_interleave_mode() != 0 || probe.scan_component_count() != 1)
                    return false;
                specs.push_back(ScanSpec{f.width, f.height, 1, 0, f.bits_per_sample, probe.parameters().near_lossless,
                                         probe.parameters().transformation, probe.validated_pc(), probe.parameters().restart_interval});
                offsets.push_back(static_cast<size_t>(probe.position() - base));
                if (c + 1 == count)
                    break;
                // the next marker that is not a restart marker ends this scan
                const uint8_t* p = probe.position();
                const uint8_t* end = p + probe.remaining();
                for (;; ++p)
                {
                    p = static_cast<const uint8_t*>(std::memchr(p, 0xFF, static_cast<size_t>(end - p)));
                    if (p == nullptr || p + 1 >= end)
                        return false;
                    if (p[1] >= 0x80 && !(p[1] >= 0xD0 && p[1] <= 0xD7))
                        break;
                }
                marker_at.push_back(static_cast<size_t>(p - base));
                probe.advance(static_cast<size_t>(p - probe.position()));
                probe.read_next_start_of_scan();
            }
            after = probe;
        }
        catch (const error&)
        {
            return false;
        }
        for (size_t c = 1; c < count; ++c)
        {
            const ScanSpec &a = specs[0], &b = specs[c];
            if (a.near_lossless != b.near_lossless || a.color_transformation != b.color_transformation ||
                a.restart_interval != b.restart_interval || std::memcmp(&a.pc, &b.pc, sizeof a.pc) != 0)
                return false;
        }
        engine.upload_stream(base, reader.remaining(), frame_hint(f.width, f.height, f.bits_per_sample));
        uploaded = true;
        std::vector<ScanResult> results(count);
        try
        {
            engine.decode_planes(specs[0], offsets.data(), static_cast<uint32_t>(count), results.data());
        }
        catch (const error&)
        {
            return false;
        }
        for (size_t c = 0; c < count; ++c)
        {
            if (results[c].errc != kOk)
                return false;
            // scan c must end exactly at the marker the probe found behind it
            if (c + 1 < count && offsets[c] + static_cast<size_t>(results[c].bytes) != marker_at[c])
                return false;
        }
        try
        { // what follows the last scan must read as the end of the image, before anything is handed to the caller
            after.advance(static_cast<size_t>(results[count - 1].bytes));
            after.read_end_of_image();
        }
        catch (const error&)
        {
            return false;
        }
        // replay the real reader over the same path (handlers fire exactly as in the scan-by-scan path), planes out; the
        // probe took every one of these steps without raising
        for (size_t c = 0; c < count; ++c)
        {
            if (static_cast<size_t>(reader.position() - base) != offsets[c])
                raise(CHARLS_JPEGLS_ERRC_INVALID_DATA); // (cannot happen: the probe took the same steps)
            engine.fetch_decoded_plane(specs[0], static_cast<uint32_t>(c), dst + stride * f.height * c, stride);
            reader.advance(static_cast<size_t>(results[c].bytes));
            if (c + 1 < count)
                reader.read_next_start_of_scan();
        }
        reader.read_end_of_image();
        state = State::completed;
        return true;
    }

    void decode(void* destination, size_t destination_size_bytes, size_t stride_arg) // reference :177-201
    {
        check_buffer(destination, destination_size_bytes);
        check_operation(state == State::header_read);
        const CallScope call(engine);
        auto* dst = static_cast<uint8_t*>(destination);
        size_t dst_left = destination_size_bytes;
        const uint8_t* base = reader.position();
        bool uploaded = false;
        if (decode_planes_together(dst, dst_left, stride_arg, base, uploaded))
            return;

        for (size_t component = 0;;)
        {
            // reference :211-236
            const size_t min_stride = minimum_stride();
            size_t stride = stride_arg;
            if (stride == 0)
                stride = min_stride;
            else if (stride < min_stride)
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
            const size_t unused = stride - min_stride;
            const uint32_t height = reader.frame_info().height;
            const size_t needed = (reader.scan_interleave_mode() == 0
                                       ? stride * reader.scan_component_count() * height
                                       : stride * height) -
                                  unused;
            if (dst_left < needed)
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);

            const charls_frame_info& f = reader.frame_info();
            const ScanSpec spec{f.width,
                                f.height,
                                static_cast<int32_t>(reader.scan_component_count()),
                                reader.parameters().interleave_mode,
                                f.bits_per_sample,
                                reader.parameters().near_lossless,
                                reader.parameters().transformation,
                                reader.validated_pc(),
                                reader.parameters().restart_interval};
            if (!uploaded)
            { // everything from the first entropy-coded byte to the end of the source goes to the device once
                engine.upload_stream(base, reader.remaining(), frame_hint(f.width, f.height, f.bits_per_sample));
                uploaded = true;
            }
            const size_t used = engine.decode_scan(spec, static_cast<size_t>(reader.position() - base), dst, stride);
            reader.advance(used);

            component += reader.scan_component_count();
            if (component == reader.component_count())
                break;
            // (the reference's span::subspan only asserts here; a wrapped size would let later planes write past the buffer)
            if (dst_left < stride * height)
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
            dst += stride * height;
            dst_left -= stride * height;
            reader.read_next_start_of_scan();
        }
        reader.read_end_of_image();
        state = State::completed;
    }

    State state{State::initial};
    StreamReader reader;
    ScanEngine engine;
};

#define JLS_THUNK_BEGIN try {
#define JLS_THUNK_END                          \
    return CHARLS_JPEGLS_ERRC_SUCCESS;         \
    }                                          \
    catch (...) { return current_exception_to_errc(); }

using D = charls_jpegls_decoder;

extern "C" {

charls_jpegls_decoder* charls_jpegls_decoder_create(void)
{
    return new (std::nothrow) charls_jpegls_decoder;
}

void charls_jpegls_decoder_destroy(const charls_jpegls_decoder* decoder)
{
    delete decoder;
}

charls_jpegls_errc charls_jpegls_decoder_set_source_buffer(charls_jpegls_decoder* d, const void* source, size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(d);
    check_buffer(source, size);
    check_operation(d->state == D::State::initial);
    d->reader.set_source(static_cast<const uint8_t*>(source), size);
    d->state = D::State::source_set;
    d->engine.expect_call(true); // (the read_header / decode_to_buffer calls of this thread follow: others about to launch wait for them)
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_read_spiff_header(charls_jpegls_decoder* d, charls_spiff_header* header,
                                                           int32_t* header_found)
{
    JLS_THUNK_BEGIN
    // C++17 evaluates the right side of the reference's `*check_pointer(header_found) = ...->read_header(...)` first:
    // the header is parsed (and the state advanced) before a null `header_found` is reported.
    check_pointer(d);
    check_pointer(header);
    check_operation(d->state == D::State::source_set);
    bool found = false;
    d->reader.read_header(header, &found);
    d->state = found ? D::State::spiff_header_read : D::State::spiff_header_not_found;
    *check_pointer(header_found) = found ? 1 : 0;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_read_header(charls_jpegls_decoder* d)
{
    JLS_THUNK_BEGIN
    check_pointer(d);
    check_operation(d->state >= D::State::source_set && d->state < D::State::header_read);
    if (d->state != D::State::spiff_header_not_found)
        d->reader.read_header();
    d->state = d->reader.end_of_image() ? D::State::completed : D::State::header_read;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_frame_info(const charls_jpegls_decoder* d, charls_frame_info* frame_info)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_header_read(); // value first, output pointer second (reference evaluation order)
    *check_pointer(frame_info) = d->reader.frame_info();
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_near_lossless(const charls_jpegls_decoder* d, int32_t component_index,
                                                           int32_t* near_lossless)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_header_read();
    check_argument(static_cast<size_t>(component_index) < d->reader.component_count());
    *check_pointer(near_lossless) = d->reader.near_lossless(static_cast<size_t>(component_index));
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_interleave_mode(const charls_jpegls_decoder* d, int32_t component_index,
                                                             charls_interleave_mode* mode)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_header_read();
    check_argument(static_cast<size_t>(component_index) < d->reader.component_count());
    *check_pointer(mode) = d->reader.interleave_mode(static_cast<size_t>(component_index));
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_preset_coding_parameters(const charls_jpegls_decoder* d, int32_t /*reserved*/,
                                                                      charls_jpegls_pc_parameters* pc)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_header_read();
    *check_pointer(pc) = d->reader.preset_coding_parameters();
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_color_transformation(const charls_jpegls_decoder* d,
                                                                  charls_color_transformation* transformation)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_header_read();
    *check_pointer(transformation) = d->reader.parameters().transformation;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_get_destination_size(const charls_jpegls_decoder* d, uint32_t stride,
                                                              size_t* size)
{
    JLS_THUNK_BEGIN
    const size_t value = check_pointer(d)->destination_size(stride);
    *check_pointer(size) = value;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_decode_to_buffer(charls_jpegls_decoder* d, void* destination, size_t size,
                                                          uint32_t stride)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->decode(destination, size, stride);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_at_comment(charls_jpegls_decoder* d, charls_at_comment_handler handler,
                                                    void* user_context)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->reader.at_comment(handler, user_context);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_decoder_at_application_data(charls_jpegls_decoder* d,
                                                             charls_at_application_data_handler handler,
                                                             void* user_context)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->reader.at_application_data(handler, user_context);
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_get_compressed_data_format(const charls_jpegls_decoder* d,
                                                             charls_compressed_data_format* format)
{
    JLS_THUNK_BEGIN
    const charls_compressed_data_format value = check_pointer(d)->reader.compressed_data_format();
    *check_pointer(format) = value;
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_get_mapping_table_id(const charls_jpegls_decoder* d, int32_t component_index,
                                                       int32_t* table_id)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_completed();
    check_argument(static_cast<size_t>(component_index) < d->reader.component_count());
    *check_pointer(table_id) = d->reader.mapping_table_id(static_cast<size_t>(component_index));
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_find_mapping_table_index(const charls_jpegls_decoder* d, int32_t mapping_table_id,
                                                           int32_t* index)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_completed();
    check_argument(mapping_table_id >= 1 && mapping_table_id <= 255);
    *check_pointer(index) = d->reader.find_mapping_table_index(static_cast<uint8_t>(mapping_table_id));
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_get_mapping_table_count(const charls_jpegls_decoder* d, int32_t* count)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_completed();
    *check_pointer(count) = static_cast<int32_t>(d->reader.mapping_table_count());
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_get_mapping_table_info(const charls_jpegls_decoder* d, int32_t index,
                                                         charls_mapping_table_info* info)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_completed();
    check_argument(static_cast<size_t>(index) < d->reader.mapping_table_count());
    *check_pointer(info) = d->reader.mapping_table_info(static_cast<size_t>(index));
    JLS_THUNK_END
}

charls_jpegls_errc charls_decoder_get_mapping_table_data(const charls_jpegls_decoder* d, int32_t index, void* data,
                                                         size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(d)->check_completed();
    check_argument(static_cast<size_t>(index) < d->reader.mapping_table_count());
    check_buffer(data, size);
    d->reader.mapping_table_data(static_cast<size_t>(index), static_cast<uint8_t*>(data), size);
    JLS_THUNK_END
}

} // extern "C"
