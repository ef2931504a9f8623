// stream_reader.h -- parses and validates the marker segments of a JPEG-LS stream (everything except entropy data).
// Accepts what the reference's src/jpeg_stream_reader.cpp:87-1012 accepts and rejects with the same error codes:
// SOI, SPIFF header + directory, COM/APPn (callbacks), APP8 "mrfx", SOF55, LSE types 1-4, DRI, DNL, SOS, EOI,
// abbreviated formats, mapping tables (data stays in the caller's buffer, referenced as fragments).
#pragma once
#include <cstring>
#include <vector>

#include "common.h"
#include "preset.h"

namespace jls {

struct CodingParameters
{
    int32_t near_lossless{};
    uint32_t restart_interval{};
    int32_t interleave_mode{};
    int32_t transformation{};
};

class StreamReader
{
public:
    void set_source(const uint8_t* data, size_t size) noexcept
    {
        pos_ = data;
        end_ = data + size;
    }

    // Continue on another window of the same stream (the batch decoder re-fetches a few KB around every marker
    // segment): mapping-table fragments point into the PREVIOUS window, so their data is let go of -- ids, entry sizes
    // and lengths stay (read_end_of_image() only looks at the ids).
    void continue_on_window(const uint8_t* data, size_t size) noexcept
    {
        for (Table& t : tables_)
            for (auto& f : t.fragments)
                f.first = nullptr;
        set_source(data, size);
    }

    void at_comment(charls_at_comment_handler h, void* ctx) noexcept
    {
        comment_handler_ = h;
        comment_ctx_ = ctx;
    }
    void at_application_data(charls_at_application_data_handler h, void* ctx) noexcept
    {
        app_handler_ = h;
        app_ctx_ = ctx;
    }

    // Reads up to and including the first SOS (or stops after a SPIFF header when the caller asked for one).
    void read_header(charls_spiff_header* header = nullptr, bool* spiff_found = nullptr);
    void read_next_start_of_scan();
    void read_end_of_image();

    const charls_frame_info& frame_info() const noexcept { return frame_; }
    const CodingParameters& parameters() const noexcept { return params_; }
    const charls_jpegls_pc_parameters& preset_coding_parameters() const noexcept { return pc_; }
    charls_jpegls_pc_parameters validated_pc() const
    {
        charls_jpegls_pc_parameters out;
        if (!pc_validate(pc_, bit_max_value(frame_.bits_per_sample), params_.near_lossless, &out))
            raise(CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_JPEGLS_PRESET_PARAMETERS);
        return out;
    }
    bool end_of_image() const noexcept { return state_ == State::after_eoi; }
    size_t component_count() const noexcept { return components_.size(); }
    uint32_t scan_component_count() const noexcept { return scan_components_; }
    int32_t scan_interleave_mode() const noexcept { return scan_ilv_; }
    int32_t near_lossless(size_t i) const noexcept { return components_[i].near; }
    int32_t interleave_mode(size_t i) const noexcept { return components_[i].ilv; }
    int32_t mapping_table_id(size_t i) const noexcept { return components_[i].table_id; }
    charls_compressed_data_format compressed_data_format() const noexcept { return format_; }

    const uint8_t* position() const noexcept { return pos_; }
    size_t remaining() const noexcept { return static_cast<size_t>(end_ - pos_); }
    void advance(size_t n) noexcept { pos_ += n; }

    size_t mapping_table_count() const noexcept { return tables_.size(); }
    int32_t find_mapping_table_index(uint8_t id) const noexcept
    {
        for (size_t i = 0; i < tables_.size(); ++i)
            if (tables_[i].id == id)
                return static_cast<int32_t>(i);
        return -1;
    }
    charls_mapping_table_info mapping_table_info(size_t i) const
    {
        return {tables_[i].id, tables_[i].entry_size, static_cast<uint32_t>(tables_[i].size())};
    }
    void mapping_table_data(size_t i, uint8_t* dst, size_t cap) const
    {
        if (tables_[i].size() > cap)
            raise(CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL);
        for (const auto& f : tables_[i].fragments)
            if (f.first == nullptr) // the source window that held the table is gone (continue_on_window)
                raise(CHARLS_JPEGLS_ERRC_INVALID_OPERATION);
        for (const auto& f : tables_[i].fragments)
        {
            std::memcpy(dst, f.first, f.second);
            dst += f.second;
        }
    }

private:
    enum class State
    {
        before_soi,
        header,
        spiff_directory,
        frame,
        scan,
        bit_stream,
        after_eoi
    };
    struct Component
    {
        uint8_t id, near, table_id;
        int32_t ilv;
    };
    struct Table
    {
        uint8_t id, entry_size;
        std::vector<std::pair<const uint8_t*, size_t>> fragments;
        size_t size() const
        {
            size_t n = 0;
            for (const auto& f : fragments)
                n += f.second;
            return n;
        }
    };

    uint8_t byte_checked()
    {
        if (pos_ == end_)
            raise(CHARLS_JPEGLS_ERRC_NEED_MORE_DATA);
        return *pos_++;
    }
    uint32_t u8() noexcept { return *pos_++; }
    uint32_t u16() noexcept
    {
        const uint32_t v = static_cast<uint32_t>(pos_[0] << 8 | pos_[1]);
        pos_ += 2;
        return v;
    }
    uint32_t u24() noexcept
    {
        const uint32_t hi = u8();
        return (hi << 16) + u16();
    }
    uint32_t u32() noexcept
    {
        const uint32_t hi = u16();
        return (hi << 16) | u16();
    }
    uint32_t next_marker();
    uint32_t marker_code();
    void validate_marker(uint32_t m) const;
    void read_segment_size();
    void need_at_least(size_t n) const
    {
        if (n > seg_size_)
            raise(CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE);
    }
    void need_exactly(size_t n) const
    {
        if (n != seg_size_)
            raise(CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE);
    }
    void skip_rest() noexcept { pos_ = seg_ + seg_size_; }
    void marker_segment(uint32_t m, charls_spiff_header* header, bool* spiff_found);
    void spiff_directory_entry(uint32_t m);
    void start_of_frame();
    void start_of_scan();
    void preset_parameters();
    void restart_interval();
    uint32_t number_of_lines();
    void app8(charls_spiff_header* header, bool* spiff_found);
    void find_number_of_lines();
    void set_height(uint32_t h, bool final_update);
    void set_width(uint32_t w);
    void call_app(uint32_t m) const;
    bool abbreviated_tables_only() const
    {
        if (tables_.empty())
            return false;
        if (state_ == State::frame)
            raise(CHARLS_JPEGLS_ERRC_ABBREVIATED_FORMAT_AND_SPIFF_HEADER_MISMATCH);
        return state_ == State::header;
    }

    const uint8_t* pos_{};
    const uint8_t* end_{};
    const uint8_t* seg_{};
    size_t seg_size_{};
    charls_frame_info frame_{};
    CodingParameters params_{};
    charls_jpegls_pc_parameters pc_{};
    std::vector<Component> components_;
    std::vector<Table> tables_;
    State state_{State::before_soi};
    uint32_t read_components_{};
    uint32_t scan_components_{};
    int32_t scan_ilv_{};
    bool dnl_expected_{};
    charls_compressed_data_format format_{};
    charls_at_comment_handler comment_handler_{};
    void* comment_ctx_{};
    charls_at_application_data_handler app_handler_{};
    void* app_ctx_{};
};

} // namespace jls
