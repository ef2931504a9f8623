// stream_writer.h -- serialises the JPEG-LS interchange format around the entropy-coded segments:
// SOI, SPIFF header + directory, COM, APPn, APP8 "mrfx", SOF55, LSE (types 1-4), SOS, EOI; all big-endian.
// Byte layout as produced by the reference's src/jpeg_stream_writer.cpp:20-243 (ISO 14495-1 annex C, T.81 annex B).
#pragma once
#include <cstring>
#include <vector>

#include "common.h"

namespace jls {

class StreamWriter
{
public:
    void set_destination(uint8_t* data, size_t size) noexcept
    {
        data_ = data;
        size_ = size;
    }
    size_t bytes_written() const noexcept { return offset_; }
    uint8_t* position() const noexcept { return data_ + offset_; }
    size_t remaining() const noexcept { return size_ - offset_; }
    void advance(size_t n) noexcept { offset_ += n; }
    void rewind() noexcept
    {
        offset_ = 0;
        component_index_ = 0;
    }
    void set_mapping_table_id(size_t component_index, int32_t table_id)
    {
        if (table_ids_.empty())
            table_ids_.resize(kMaxComponents);
        table_ids_[component_index] = static_cast<uint8_t>(table_id);
    }

    void start_of_image() { marker_only(0xD8); }

    void end_of_image(bool even_size)
    {
        if (even_size && (offset_ % 2) != 0)
        {
            need(1);
            put8(0xFF); // fill byte before the marker keeps the total even
        }
        marker_only(0xD9);
    }

    void spiff_header(const charls_spiff_header& h)
    {
        segment(0xE8, 30);
        static const uint8_t magic[6] = {'S', 'P', 'I', 'F', 'F', 0};
        put(magic, 6);
        put8(2); // version 2.0
        put8(0);
        put8(static_cast<uint32_t>(h.profile_id));
        put8(static_cast<uint32_t>(h.component_count));
        put32(h.height);
        put32(h.width);
        put8(static_cast<uint32_t>(h.color_space));
        put8(static_cast<uint32_t>(h.bits_per_sample));
        put8(static_cast<uint32_t>(h.compression_type));
        put8(static_cast<uint32_t>(h.resolution_units));
        put32(h.vertical_resolution);
        put32(h.horizontal_resolution);
    }

    void spiff_directory_entry(uint32_t tag, const uint8_t* data, size_t size)
    {
        segment(0xE8, 4 + size);
        put32(tag);
        put(data, size);
    }

    void spiff_end_of_directory()
    { // EOD entry (type 1) whose last two data bytes are the SOI of the embedded stream
        static const uint8_t eod[6] = {0, 0, 0, 1, 0xFF, 0xD8};
        segment(0xE8, 6);
        put(eod, 6);
    }

    void color_transform(int32_t transformation)
    {
        segment(0xE8, 5);
        put(reinterpret_cast<const uint8_t*>("mrfx"), 4);
        put8(static_cast<uint32_t>(transformation));
    }

    void comment(const uint8_t* data, size_t size)
    {
        segment(0xFE, size);
        put(data, size);
    }

    void application_data(int32_t id, const uint8_t* data, size_t size)
    {
        segment(0xE0 + static_cast<uint32_t>(id), size);
        put(data, size);
    }

    // returns true when the dimensions do not fit SOF and must follow in an LSE type 4 segment
    bool start_of_frame(const charls_frame_info& f)
    {
        segment(0xF7, 6 + static_cast<size_t>(f.component_count) * 3);
        const bool oversize = f.width > 65535 || f.height > 65535;
        put8(static_cast<uint32_t>(f.bits_per_sample));
        put16(oversize ? 0 : f.height);
        put16(oversize ? 0 : f.width);
        put8(static_cast<uint32_t>(f.component_count));
        for (int32_t id = 1; id <= f.component_count; ++id)
        {
            put8(static_cast<uint32_t>(id));
            put8(0x11);
            put8(0);
        }
        return oversize;
    }

    void preset_coding_parameters(const charls_jpegls_pc_parameters& p)
    {
        segment(0xF8, 11);
        put8(1);
        put16(static_cast<uint32_t>(p.maximum_sample_value));
        put16(static_cast<uint32_t>(p.threshold1));
        put16(static_cast<uint32_t>(p.threshold2));
        put16(static_cast<uint32_t>(p.threshold3));
        put16(static_cast<uint32_t>(p.reset_value));
    }

    void oversize_dimensions(uint32_t height, uint32_t width)
    {
        segment(0xF8, 10);
        put8(4);
        put8(4); // Wxy: always 4 bytes per dimension
        put32(height);
        put32(width);
    }

    void mapping_table(int32_t table_id, int32_t entry_size, const uint8_t* data, size_t size)
    { // first LSE type 2, remainder as LSE type 3 continuation segments
        const size_t max_chunk = kSegmentMaxData - 3;
        size_t done = 0;
        uint32_t type = 2;
        do
        {
            const size_t n = std::min(size - done, max_chunk);
            segment(0xF8, 3 + n);
            put8(type);
            put8(static_cast<uint32_t>(table_id));
            put8(static_cast<uint32_t>(entry_size));
            put(data + done, n);
            done += n;
            type = 3;
        } while (done < size);
    }

    // DRI segment as the reference's reader accepts it (src/jpeg_stream_reader.cpp:586-607): 16-bit interval when it
    // fits, else 32-bit.  The reference's writer never emits it (restart-interval encoding is an extension here).
    void define_restart_interval(uint32_t lines)
    {
        if (lines <= 65535)
        {
            segment(0xDD, 2);
            put16(lines);
        }
        else
        {
            segment(0xDD, 4);
            put32(lines);
        }
    }

    void start_of_scan(int32_t component_count, int32_t near, int32_t ilv)
    {
        segment(0xDA, 1 + static_cast<size_t>(component_count) * 2 + 3);
        put8(static_cast<uint32_t>(component_count));
        for (int32_t i = 0; i < component_count; ++i)
        {
            put8(component_index_ + 1u);
            put8(table_ids_.empty() ? 0u : table_ids_[component_index_]);
            ++component_index_;
        }
        put8(static_cast<uint32_t>(near));
        put8(static_cast<uint32_t>(ilv));
        put8(0);
    }

private:
    void need(size_t n) const
    {
        if (offset_ + n > size_)
            raise(CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL);
    }
    void marker_only(uint32_t code)
    {
        need(2);
        put8(0xFF);
        put8(code);
    }
    void segment(uint32_t code, size_t data_size)
    { // the whole segment must fit before anything is written
        need(4 + data_size);
        put8(0xFF);
        put8(code);
        put16(static_cast<uint32_t>(2 + data_size));
    }
    void put8(uint32_t v) noexcept { data_[offset_++] = static_cast<uint8_t>(v); }
    void put16(uint32_t v) noexcept
    {
        put8(v >> 8);
        put8(v);
    }
    void put32(uint32_t v) noexcept
    {
        put16(v >> 16);
        put16(v & 0xFFFFu);
    }
    void put(const uint8_t* p, size_t n) noexcept
    {
        if (n)
            std::memcpy(data_ + offset_, p, n);
        offset_ += n;
    }

    uint8_t* data_{};
    size_t size_{};
    size_t offset_{};
    uint8_t component_index_{};
    std::vector<uint8_t> table_ids_;
};

} // namespace jls
