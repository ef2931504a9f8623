// encoder_api.cpp -- `struct charls_jpegls_encoder` and its 23 extern "C" entry points.
//
// Same state machine, argument checks, marker emission order and error codes as the reference facade
// (src/charls_jpegls_encoder.cpp:31-744); the one difference is where the scan is coded: every
// `make_scan_codec<scan_encoder>()->encode_scan(...)` of the reference (:285-296) is a ScanEngine call that runs the
// gfx950 kernels.
#include <cstring>
#include <new>

#include "common.h"
#include "preset.h"
#include <vector>

#include "scan_engine.h"
#include "stream_writer.h"

using namespace jls;

struct charls_jpegls_encoder
{
    enum class State
    {
        initial,
        destination_set,
        spiff_header,
        tables_and_miscellaneous,
        completed
    };

    void set_destination(void* data, size_t size) // reference :33-41
    {
        check_buffer(data, size);
        check_operation(state <= State::destination_set);
        writer.set_destination(static_cast<uint8_t*>(data), size);
        state = State::destination_set;
    }

    void set_frame_info(const charls_frame_info& f) // reference :43-53
    {
        check_argument(f.width >= 1 && f.width <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_WIDTH);
        check_argument(f.height >= 1 && f.height <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_HEIGHT);
        check_argument(f.bits_per_sample >= kMinBits && f.bits_per_sample <= kMaxBits,
                       CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_BITS_PER_SAMPLE);
        check_argument(f.component_count >= 1 && f.component_count <= kMaxComponents,
                       CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT);
        frame = f;
        // (the encode call of this thread follows: others about to launch a batch of this geometry wait a moment for it)
        engine.expect_call(false, frame_hint(f.width, f.height, f.bits_per_sample));
    }

    bool frame_configured() const noexcept { return frame.width != 0; }
    bool has_option(uint32_t o) const noexcept { return (options & o) == o; }
    void check_can_write() const { check_operation(state >= State::destination_set && state < State::completed); }

    size_t estimated_destination_size() const // reference :103-114
    {
        check_operation(frame_configured());
        size_t size = checked_mul(checked_mul(checked_mul(frame.width, frame.height),
                                              static_cast<size_t>(frame.component_count)),
                                  bytes_per_sample(frame.bits_per_sample));
        size_t extra = size / 16 + 1024 + kSpiffHeaderSize;
        if (restart_interval != 0) // extension: DRI segment + two marker bytes per interval and scan
            extra += 6 + 2 * static_cast<size_t>((frame.height + restart_interval - 1) / restart_interval) *
                             static_cast<size_t>(frame.component_count);
        return size + extra < size ? SIZE_MAX : size + extra;
    }

    void to_tables_and_misc() // reference :360-386
    {
        if (state == State::tables_and_miscellaneous)
            return;
        if (state == State::spiff_header)
            writer.spiff_end_of_directory();
        else
            writer.start_of_image();
        if (has_option(2))
        {
            static const char version[] = "charls 3.0.0"; // written with its terminating NUL, as the reference does
            writer.comment(reinterpret_cast<const uint8_t*>(version), sizeof version);
        }
        state = State::tables_and_miscellaneous;
    }

    void write_spiff_header_core(const charls_spiff_header& h) // reference :276-283
    {
        check_operation(state == State::destination_set);
        writer.start_of_image();
        writer.spiff_header(h);
        state = State::spiff_header;
    }

    void write_end_of_image() // reference :420-424
    {
        writer.end_of_image(has_option(1));
        state = State::completed;
    }

    size_t minimum_stride(int32_t source_components) const noexcept // reference :323-331
    {
        const size_t s = static_cast<size_t>(frame.width) * bytes_per_sample(frame.bits_per_sample);
        return interleave == 0 ? s : s * static_cast<size_t>(source_components);
    }

    void encode_components(const void* source, size_t source_size, int32_t source_components, size_t stride)
    { // reference :182-236
        check_buffer(source, source_size);
        check_can_write();
        check_operation(frame_configured());
        if (frame.component_count == 1 && interleave != 0)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_INTERLEAVE_MODE);
        const int32_t bit_maxval = bit_max_value(frame.bits_per_sample);
        { // reference :344-358
            int32_t maxval = bit_maxval;
            if (user_pc.maximum_sample_value != 0)
            {
                if (user_pc.maximum_sample_value < 1 || user_pc.maximum_sample_value > bit_maxval)
                    raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_JPEGLS_PC_PARAMETERS);
                maxval = user_pc.maximum_sample_value;
            }
            if (near > max_near_for(maxval))
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_NEAR_LOSSLESS);
        }
        // reference :298-321
        const size_t min_stride = minimum_stride(source_components);
        if (stride == 0)
            stride = min_stride;
        else if (stride < min_stride)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE);
        const size_t unused = stride - min_stride;
        const size_t min_size = (interleave == 0
                                     ? checked_mul(stride * static_cast<size_t>(source_components), frame.height)
                                     : checked_mul(stride, frame.height)) -
                                unused;
        if (source_size < min_size)
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);

        if (!pc_validate(user_pc, bit_maxval, near, &pc))
            raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_JPEGLS_PC_PARAMETERS);

        if (encoded_components == 0)
        {
            to_tables_and_misc();
            if (transformation != 0) // reference :388-398
            {
                if (!color_transformation_possible(frame, near, interleave))
                    raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COLOR_TRANSFORMATION);
                writer.color_transform(transformation);
            }
            if (writer.start_of_frame(frame)) // reference :400-407
                writer.oversize_dimensions(frame.height, frame.width);
            if (!pc_is_default(user_pc, default_pc(bit_maxval, near)) || (has_option(4) && frame.bits_per_sample > 12))
                writer.preset_coding_parameters(pc); // reference :409-418
            if (restart_interval != 0)
                writer.define_restart_interval(restart_interval); // extension: charls_amd_jpegls_encoder_set_restart_interval
        }

        const CallScope call(engine);
        engine.upload_pixels(static_cast<const uint8_t*>(source), min_size, frame_hint(frame.width, frame.height, frame.bits_per_sample));
        ScanSpec spec{frame.width, frame.height, 1, interleave, frame.bits_per_sample, near, transformation, pc, restart_interval};
        if (interleave == 0)
        {
            const size_t plane_bytes = stride * frame.height;
            // The component scans of a planar frame share nothing (own contexts, own bit stream): all of them are coded by
            // one launch into private buffers and then placed one after the other, where the reference codes them in a
            // loop (:209-224).  What the reference's scan c would have been given as its destination -- everything behind
            // its own SOS header -- is only known once scans 0..c-1 have been placed: a scan that, placed, leaves fewer
            // than 4 bytes (or does not fit) is coded again on its own with exactly that destination, so that
            // destination_too_small is raised by the same rule as before (src/scan_encoder.hpp:117-120).
            std::vector<ScanResult> planes(static_cast<size_t>(source_components));
            const bool together = source_components > 1;
            if (together)
                engine.encode_planes(spec, static_cast<uint32_t>(source_components), plane_bytes, stride, writer.remaining(), planes.data());
            for (int32_t c = 0; c < source_components; ++c)
            {
                writer.start_of_scan(1, near, interleave);
                const ScanResult& r = planes[static_cast<size_t>(c)];
                size_t n;
                if (together && r.errc == kOk && r.bytes + 4 <= writer.remaining())
                {
                    n = static_cast<size_t>(r.bytes);
                    engine.fetch_encoded_scan(static_cast<uint32_t>(c), writer.position(), n);
                }
                else
                    n = engine.encode_scan(spec, plane_bytes * static_cast<size_t>(c), stride, writer.position(), writer.remaining());
                writer.advance(n);
            }
        }
        else
        {
            if (source_components < 2 || source_components > kMaxComponentsInScan)
                raise(CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT); // the reference asserts here
            writer.start_of_scan(source_components, near, interleave);
            spec.components = source_components;
            const size_t n = engine.encode_scan(spec, 0, stride, writer.position(), writer.remaining());
            writer.advance(n);
        }

        encoded_components += source_components;
        if (encoded_components == frame.component_count)
            write_end_of_image();
    }

    charls_frame_info frame{};
    int32_t near{};
    int32_t encoded_components{};
    int32_t interleave{};
    int32_t transformation{};
    uint32_t options{};
    uint32_t restart_interval{}; // lines; 0 = none (extension)
    State state{State::initial};
    StreamWriter writer;
    charls_jpegls_pc_parameters user_pc{};
    charls_jpegls_pc_parameters pc{};
    ScanEngine engine;
};

#define JLS_THUNK_BEGIN try {
#define JLS_THUNK_END                          \
    return CHARLS_JPEGLS_ERRC_SUCCESS;         \
    }                                          \
    catch (...) { return current_exception_to_errc(); }

extern "C" {

charls_jpegls_encoder* charls_jpegls_encoder_create(void)
{
    return new (std::nothrow) charls_jpegls_encoder;
}

void charls_jpegls_encoder_destroy(const charls_jpegls_encoder* encoder)
{
    delete encoder;
}

charls_jpegls_errc charls_jpegls_encoder_set_destination_buffer(charls_jpegls_encoder* e, void* buffer, size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(e)->set_destination(buffer, size);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_frame_info(charls_jpegls_encoder* e, const charls_frame_info* frame_info)
{
    JLS_THUNK_BEGIN
    check_pointer(e)->set_frame_info(*check_pointer(frame_info));
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_near_lossless(charls_jpegls_encoder* e, int32_t near_lossless)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(near_lossless >= 0 && near_lossless <= kMaxNear, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_NEAR_LOSSLESS);
    e->near = near_lossless;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_encoding_options(charls_jpegls_encoder* e, charls_encoding_options options)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(options <= 7u, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_ENCODING_OPTIONS);
    e->options = options;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_interleave_mode(charls_jpegls_encoder* e, charls_interleave_mode mode)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(mode >= 0 && mode <= 2, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_INTERLEAVE_MODE);
    e->interleave = mode;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_preset_coding_parameters(charls_jpegls_encoder* e,
                                                                      const charls_jpegls_pc_parameters* p)
{
    JLS_THUNK_BEGIN
    check_pointer(e)->user_pc = *check_pointer(p); // validated when encoding starts
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_color_transformation(charls_jpegls_encoder* e,
                                                                  charls_color_transformation t)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(t >= 0 && t <= 3, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COLOR_TRANSFORMATION);
    e->transformation = t;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_set_mapping_table_id(charls_jpegls_encoder* e, int32_t component_index,
                                                              int32_t table_id)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(component_index >= 0 && component_index <= kMaxComponents - 1);
    check_argument(table_id >= 0 && table_id <= 255);
    e->writer.set_mapping_table_id(static_cast<size_t>(component_index), table_id);
    JLS_THUNK_END
}

// Extension (include/charls_amd.h part 2): restart interval in lines, 0 = none.  The reference's encoder has no such
// setter; its decoder handles the resulting DRI/RSTm stream (src/jpeg_stream_reader.cpp:586-607, src/scan_decoder.hpp:335-349).
charls_jpegls_errc charls_amd_jpegls_encoder_set_restart_interval(charls_jpegls_encoder* e, uint32_t lines)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_operation(e->encoded_components == 0 && e->state < charls_jpegls_encoder::State::completed);
    e->restart_interval = lines;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_get_estimated_destination_size(const charls_jpegls_encoder* e, size_t* size)
{
    JLS_THUNK_BEGIN
    const size_t value = check_pointer(e)->estimated_destination_size(); // value first, then the output pointer
    *check_pointer(size) = value;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_standard_spiff_header(charls_jpegls_encoder* e,
                                                                     charls_spiff_color_space color_space,
                                                                     charls_spiff_resolution_units units,
                                                                     uint32_t vertical, uint32_t horizontal)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_operation(e->frame_configured());
    e->write_spiff_header_core({0, e->frame.component_count, e->frame.height, e->frame.width, color_space,
                                e->frame.bits_per_sample, 6 /* jpeg_ls */, units, vertical, horizontal});
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_spiff_header(charls_jpegls_encoder* e, const charls_spiff_header* h)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_pointer(h);
    check_argument(h->height >= 1 && h->height <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_HEIGHT);
    check_argument(h->width >= 1 && h->width <= kMaxDimension, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_WIDTH);
    e->write_spiff_header_core(*h);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_spiff_entry(charls_jpegls_encoder* e, uint32_t entry_tag,
                                                           const void* data, size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_buffer(data, size);
    check_argument(entry_tag != 1); // the end-of-directory tag is written by the encoder itself
    check_argument(size <= kSpiffEntryMaxData, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    check_operation(e->state == charls_jpegls_encoder::State::spiff_header);
    e->writer.spiff_directory_entry(entry_tag, static_cast<const uint8_t*>(data), size);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_spiff_end_of_directory_entry(charls_jpegls_encoder* e)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_operation(e->state == charls_jpegls_encoder::State::spiff_header);
    e->to_tables_and_misc();
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_comment(charls_jpegls_encoder* e, const void* comment, size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_buffer(comment, size);
    check_argument(size <= kSegmentMaxData, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    e->check_can_write();
    e->to_tables_and_misc();
    e->writer.comment(static_cast<const uint8_t*>(comment), size);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_application_data(charls_jpegls_encoder* e, int32_t id, const void* data,
                                                                size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(id >= 0 && id <= 15);
    check_buffer(data, size);
    check_argument(size <= kSegmentMaxData, CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    e->check_can_write();
    e->to_tables_and_misc();
    e->writer.application_data(id, static_cast<const uint8_t*>(data), size);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_write_mapping_table(charls_jpegls_encoder* e, int32_t table_id,
                                                             int32_t entry_size, const void* data, size_t size)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_argument(table_id >= 1 && table_id <= 255);
    check_argument(entry_size >= 1 && entry_size <= 255);
    check_buffer(data, size);
    check_argument(size >= static_cast<size_t>(entry_size), CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE);
    e->check_can_write();
    e->to_tables_and_misc();
    e->writer.mapping_table(table_id, entry_size, static_cast<const uint8_t*>(data), size);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_encode_from_buffer(charls_jpegls_encoder* e, const void* source,
                                                            size_t source_size, uint32_t stride)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    e->encode_components(source, source_size, e->frame.component_count, stride);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_encode_components_from_buffer(charls_jpegls_encoder* e, const void* source,
                                                                       size_t source_size, int32_t source_components,
                                                                       uint32_t stride)
{
    JLS_THUNK_BEGIN
    check_pointer(e)->encode_components(source, source_size, source_components, stride);
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_create_abbreviated_format(charls_jpegls_encoder* e)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    check_operation(e->state == charls_jpegls_encoder::State::tables_and_miscellaneous);
    e->write_end_of_image();
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_get_bytes_written(const charls_jpegls_encoder* e, size_t* bytes_written)
{
    JLS_THUNK_BEGIN
    const size_t value = check_pointer(e)->writer.bytes_written();
    *check_pointer(bytes_written) = value;
    JLS_THUNK_END
}

charls_jpegls_errc charls_jpegls_encoder_rewind(charls_jpegls_encoder* e)
{
    JLS_THUNK_BEGIN
    check_pointer(e);
    if (e->state != charls_jpegls_encoder::State::initial)
    { // reference :250-258
        e->writer.rewind();
        e->state = charls_jpegls_encoder::State::destination_set;
        e->encoded_components = 0;
    }
    JLS_THUNK_END
}

} // extern "C"
