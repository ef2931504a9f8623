// stream_reader.cpp -- see stream_reader.h.  Error behaviour follows reference src/jpeg_stream_reader.cpp (cited per block).
#include "stream_reader.h"

namespace jls {

namespace {
constexpr uint32_t kSOI = 0xD8, kEOI = 0xD9, kSOS = 0xDA, kDNL = 0xDC, kDRI = 0xDD, kAPP0 = 0xE0, kAPP8 = 0xE8,
                   kAPP15 = 0xEF, kCOM = 0xFE, kSOF55 = 0xF7, kLSE = 0xF8;

bool other_jpeg_sof(uint32_t m) // reference :28-60
{
    switch (m)
    {
    case 0xC0: case 0xC1: case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xF9:
        return true;
    default:
        return false;
    }
}
} // namespace

void StreamReader::read_header(charls_spiff_header* header, bool* spiff_found) // reference :87-149
{
    if (state_ == State::before_soi)
    {
        if (next_marker() != kSOI)
            raise(CHARLS_JPEGLS_ERRC_START_OF_IMAGE_MARKER_NOT_FOUND);
        components_.reserve(4);
        state_ = State::header;
    }
    for (;;)
    {
        const uint32_t m = next_marker();
        if (m == kEOI)
        {
            if (abbreviated_tables_only())
            {
                state_ = State::after_eoi;
                format_ = 3; // abbreviated_table_specification
                return;
            }
            raise(CHARLS_JPEGLS_ERRC_UNEXPECTED_END_OF_IMAGE_MARKER);
        }
        validate_marker(m);
        read_segment_size();
        if (state_ == State::spiff_directory)
            spiff_directory_entry(m);
        else
            marker_segment(m, header, spiff_found);

        if (state_ == State::header && spiff_found && *spiff_found)
        {
            state_ = State::spiff_directory;
            return;
        }
        if (state_ == State::bit_stream)
        {
            if (frame_.height == 0)
                find_number_of_lines();
            if (frame_.width < 1)
                raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_WIDTH);
            if (params_.transformation != 0 &&
                !color_transformation_possible(frame_, near_lossless(0), interleave_mode(0)))
                raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COLOR_TRANSFORMATION);
            return;
        }
    }
}

void StreamReader::read_end_of_image() // reference :152-173 (tolerates one zero padding byte before EOI)
{
    uint32_t b = byte_checked();
    if (b == 0)
        b = byte_checked();
    if (b != 0xFF || marker_code() != kEOI)
        raise(CHARLS_JPEGLS_ERRC_END_OF_IMAGE_MARKER_NOT_FOUND);
    bool external_tables = false;
    for (const auto& c : components_)
        if (c.table_id != 0 && find_mapping_table_index(c.table_id) < 0)
            external_tables = true;
    format_ = external_tables ? 2 : 1;
    state_ = State::after_eoi;
}

void StreamReader::read_next_start_of_scan() // reference :176-189
{
    state_ = State::scan;
    do
    {
        const uint32_t m = next_marker();
        validate_marker(m);
        read_segment_size();
        marker_segment(m, nullptr, nullptr);
    } while (state_ == State::scan);
}

uint32_t StreamReader::next_marker() // reference :192-198
{
    if (byte_checked() != 0xFF)
        raise(CHARLS_JPEGLS_ERRC_JPEG_MARKER_START_BYTE_NOT_FOUND);
    return marker_code();
}

uint32_t StreamReader::marker_code() // reference :201-213: 0xFF fill bytes may precede the code
{
    uint32_t m = byte_checked();
    while (m == 0xFF)
        m = byte_checked();
    return m;
}

void StreamReader::validate_marker(uint32_t m) const // reference :216-281
{
    if (m == kSOS)
    {
        if (state_ != State::scan)
            raise(CHARLS_JPEGLS_ERRC_UNEXPECTED_START_OF_SCAN_MARKER);
        return;
    }
    if (m == kSOF55)
    {
        if (state_ == State::scan)
            raise(CHARLS_JPEGLS_ERRC_DUPLICATE_START_OF_FRAME_MARKER);
        return;
    }
    if (m == kDRI || m == kLSE || m == kCOM || (m >= kAPP0 && m <= kAPP15))
        return;
    if (m == kDNL)
    {
        if (!dnl_expected_)
            raise(CHARLS_JPEGLS_ERRC_UNEXPECTED_DEFINE_NUMBER_OF_LINES_MARKER);
        return;
    }
    if (m == kSOI)
        raise(CHARLS_JPEGLS_ERRC_DUPLICATE_START_OF_IMAGE_MARKER);
    if (other_jpeg_sof(m))
        raise(CHARLS_JPEGLS_ERRC_ENCODING_NOT_SUPPORTED);
    if (m >= 0xD0 && m <= 0xD7)
        raise(CHARLS_JPEGLS_ERRC_UNEXPECTED_RESTART_MARKER);
    raise(CHARLS_JPEGLS_ERRC_UNKNOWN_JPEG_MARKER_FOUND);
}

void StreamReader::read_segment_size() // reference :717-725
{
    if (pos_ + 2 > end_)
        raise(CHARLS_JPEGLS_ERRC_NEED_MORE_DATA);
    const size_t size = u16();
    if (size < 2 || pos_ + (size - 2) > end_)
        raise(CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE);
    seg_ = pos_;
    seg_size_ = size - 2;
}

void StreamReader::marker_segment(uint32_t m, charls_spiff_header* header, bool* spiff_found) // reference :339-392
{
    switch (m)
    {
    case kSOF55:
        start_of_frame();
        break;
    case kSOS:
        start_of_scan();
        break;
    case kLSE:
        preset_parameters();
        break;
    case kDRI:
        restart_interval();
        break;
    case kDNL:
        (void)number_of_lines();
        dnl_expected_ = false;
        break;
    case kAPP8:
        app8(header, spiff_found);
        break;
    case kCOM:
        if (comment_handler_ && comment_handler_(seg_size_ ? pos_ : nullptr, seg_size_, comment_ctx_) != 0)
            raise(CHARLS_JPEGLS_ERRC_CALLBACK_FAILED);
        skip_rest();
        break;
    default: // APP0-7, APP9-15
        call_app(m);
        skip_rest();
        break;
    }
}

void StreamReader::call_app(uint32_t m) const // reference :922-930
{
    if (app_handler_ &&
        app_handler_(static_cast<int32_t>(m - kAPP0), seg_size_ ? pos_ : nullptr, seg_size_, app_ctx_) != 0)
        raise(CHARLS_JPEGLS_ERRC_CALLBACK_FAILED);
}

void StreamReader::spiff_directory_entry(uint32_t m) // reference :394-408
{
    if (m != kAPP8)
        raise(CHARLS_JPEGLS_ERRC_MISSING_END_OF_SPIFF_DIRECTORY);
    need_at_least(4);
    if (u32() == 1) // end-of-directory entry, followed by the embedded SOI as data
    {
        need_exactly(6);
        state_ = State::frame;
    }
    skip_rest();
}

void StreamReader::start_of_frame() // reference :411-445
{
    need_at_least(6);
    frame_.bits_per_sample = static_cast<int32_t>(u8());
    if (frame_.bits_per_sample < kMinBits || frame_.bits_per_sample > kMaxBits)
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_BITS_PER_SAMPLE);
    set_height(u16(), false);
    set_width(u16());
    frame_.component_count = static_cast<int32_t>(u8());
    if (frame_.component_count == 0)
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COMPONENT_COUNT);
    need_exactly(static_cast<size_t>(frame_.component_count) * 3 + 6);
    for (int32_t i = 0; i < frame_.component_count; ++i)
    {
        const uint8_t id = static_cast<uint8_t>(u8());
        for (const auto& c : components_)
            if (c.id == id)
                raise(CHARLS_JPEGLS_ERRC_DUPLICATE_COMPONENT_ID_IN_SOF_SEGMENT);
        components_.push_back({id, 0, 0, 0});
        if (u8() != 0x11)
            raise(CHARLS_JPEGLS_ERRC_PARAMETER_VALUE_NOT_SUPPORTED);
        (void)u8();
    }
    state_ = State::scan;
}

uint32_t StreamReader::number_of_lines() // reference :464-482
{
    switch (seg_size_)
    {
    case 2:
        return u16();
    case 3:
        return u24();
    case 4:
        return u32();
    default:
        raise(CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE);
    }
}

void StreamReader::preset_parameters() // reference :485-583
{
    need_at_least(1);
    const uint32_t type = u8();
    switch (type)
    {
    case 1:
        need_exactly(11);
        pc_.maximum_sample_value = static_cast<int32_t>(u16());
        pc_.threshold1 = static_cast<int32_t>(u16());
        pc_.threshold2 = static_cast<int32_t>(u16());
        pc_.threshold3 = static_cast<int32_t>(u16());
        pc_.reset_value = static_cast<int32_t>(u16());
        return;
    case 2:
    case 3: {
        need_at_least(3);
        const uint8_t id = static_cast<uint8_t>(u8());
        const uint8_t entry_size = static_cast<uint8_t>(u8());
        const std::pair<const uint8_t*, size_t> fragment{seg_ + 3, seg_size_ - 3};
        const int32_t index = find_mapping_table_index(id);
        if (type == 2)
        {
            if (id == 0 || index >= 0)
                raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_MAPPING_TABLE_ID);
            tables_.push_back({id, entry_size, {fragment}});
        }
        else
        {
            if (index < 0 || tables_[static_cast<size_t>(index)].entry_size != entry_size)
                raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_MAPPING_TABLE_CONTINUATION);
            tables_[static_cast<size_t>(index)].fragments.push_back(fragment);
        }
        skip_rest();
        return;
    }
    case 4: {
        need_at_least(2);
        const uint32_t bytes = u8();
        uint32_t h, w;
        switch (bytes)
        {
        case 2:
            need_exactly(2 + 4);
            h = u16();
            w = u16();
            break;
        case 3:
            need_exactly(2 + 6);
            h = u24();
            w = u24();
            break;
        case 4:
            need_exactly(2 + 8);
            h = u32();
            w = u32();
            break;
        default:
            raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_JPEGLS_PRESET_PARAMETERS);
        }
        set_height(h, false);
        set_width(w);
        return;
    }
    default:
        raise(type <= 0xD ? CHARLS_JPEGLS_ERRC_JPEGLS_PRESET_EXTENDED_PARAMETER_TYPE_NOT_SUPPORTED
                          : CHARLS_JPEGLS_ERRC_INVALID_JPEGLS_PRESET_PARAMETER_TYPE);
    }
}

void StreamReader::restart_interval() // reference :586-607
{
    switch (seg_size_)
    {
    case 2:
        params_.restart_interval = u16();
        break;
    case 3:
        params_.restart_interval = u24();
        break;
    case 4:
        params_.restart_interval = u32();
        break;
    default:
        raise(CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE);
    }
}

void StreamReader::start_of_scan() // reference :610-654
{
    need_at_least(1);
    const uint32_t n = u8();
    if (n < 1 || n > static_cast<uint32_t>(kMaxComponentsInScan) ||
        n > static_cast<uint32_t>(frame_.component_count) - read_components_)
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COMPONENT_COUNT);
    scan_components_ = n;
    read_components_ += n;
    need_exactly(n * size_t{2} + 4);
    uint8_t ids[4], table_ids[4];
    for (uint32_t i = 0; i < n; ++i)
    {
        ids[i] = static_cast<uint8_t>(u8());
        table_ids[i] = static_cast<uint8_t>(u8());
    }
    params_.near_lossless = static_cast<int32_t>(u8());
    const int32_t maxval = pc_.maximum_sample_value != 0 ? pc_.maximum_sample_value : bit_max_value(frame_.bits_per_sample);
    if (params_.near_lossless > max_near_for(maxval))
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_NEAR_LOSSLESS);
    const uint32_t ilv = u8();
    if (ilv > 2 || (n == 1 && ilv != 0))
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_INTERLEAVE_MODE);
    scan_ilv_ = static_cast<int32_t>(ilv);
    params_.interleave_mode = scan_ilv_;
    for (uint32_t i = 0; i < n; ++i)
    { // reference :974-990: only non-default values are recorded against the component id
        if (table_ids[i] == 0 && params_.near_lossless == 0 && scan_ilv_ == 0)
            continue;
        Component* found = nullptr;
        for (auto& c : components_)
            if (c.id == ids[i])
            {
                found = &c;
                break;
            }
        if (!found)
            raise(CHARLS_JPEGLS_ERRC_UNKNOWN_COMPONENT_ID);
        found->near = static_cast<uint8_t>(params_.near_lossless);
        found->table_id = table_ids[i];
        found->ilv = scan_ilv_;
    }
    if ((u8() & 0xFu) != 0) // Al: point transform is not supported
        raise(CHARLS_JPEGLS_ERRC_PARAMETER_VALUE_NOT_SUPPORTED);
    state_ = State::bit_stream;
}

void StreamReader::app8(charls_spiff_header* header, bool* spiff_found) // reference :745-840
{
    call_app(kAPP8);
    if (spiff_found)
        *spiff_found = false;
    if (seg_size_ == 5)
    {
        if (std::memcmp(pos_, "mrfx", 4) == 0)
        {
            const uint32_t t = pos_[4];
            if (t <= 3)
                params_.transformation = static_cast<int32_t>(t);
            else if (t == 4 || t == 5)
                raise(CHARLS_JPEGLS_ERRC_COLOR_TRANSFORM_NOT_SUPPORTED);
            else
                raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COLOR_TRANSFORMATION);
        }
    }
    else if (header && spiff_found && seg_size_ >= 30)
    {
        static const uint8_t magic[6] = {'S', 'P', 'I', 'F', 'F', 0};
        if (std::memcmp(pos_, magic, 6) != 0 || pos_[6] > 2)
        {
            *header = {};
            *spiff_found = false;
        }
        else
        {
            pos_ += 8;
            header->profile_id = static_cast<int32_t>(u8());
            header->component_count = static_cast<int32_t>(u8());
            header->height = u32();
            header->width = u32();
            header->color_space = static_cast<int32_t>(u8());
            header->bits_per_sample = static_cast<int32_t>(u8());
            header->compression_type = static_cast<int32_t>(u8());
            header->resolution_units = static_cast<int32_t>(u8());
            header->vertical_resolution = u32();
            header->horizontal_resolution = u32();
            *spiff_found = true;
        }
    }
    skip_rest();
}

void StreamReader::find_number_of_lines() // reference :933-959: DNL must directly follow the first scan
{
    for (const uint8_t* p = pos_; p + 1 < end_; ++p)
    {
        if (*p != 0xFF)
            continue;
        const uint8_t code = p[1];
        if (code < 128 || code == 0xFF)
            continue;
        if (code != kDNL)
            break;
        const uint8_t* saved = pos_;
        pos_ = p + 2;
        read_segment_size();
        set_height(number_of_lines(), true);
        dnl_expected_ = true;
        pos_ = saved;
        return;
    }
    raise(CHARLS_JPEGLS_ERRC_DEFINE_NUMBER_OF_LINES_MARKER_NOT_FOUND);
}

void StreamReader::set_height(uint32_t h, bool final_update) // reference :898-907
{
    if (h == 0 && !final_update)
        return;
    if (frame_.height != 0 || h < 1 || h > kMaxDimension)
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_HEIGHT);
    frame_.height = h;
}

void StreamReader::set_width(uint32_t w) // reference :910-919
{
    if (w == 0)
        return;
    if (frame_.width != 0 || w > kMaxDimension)
        raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_WIDTH);
    frame_.width = w;
}

} // namespace jls
