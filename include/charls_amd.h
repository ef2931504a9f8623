/*
 * charls_amd.h -- C ABI of the MI355X-native JPEG-LS engine.
 *
 * Part 1 is the drop-in boundary: the 48 entry points, enums and four POD structs of team-charls/charls 3.0, with the
 * same names, argument meaning, state machines and error codes, so that a program built against the reference's own
 * <charls/charls.h> runs unchanged when libcharls_amd.so is loaded in place of libcharls.so.3.  Each declaration cites
 * the reference interface it replaces (paths relative to the reference tree).  Source/destination buffers passed to
 * part 1 are HOST memory, borrowed for the duration of the call exactly as in the reference.
 *
 * Part 2 (prefix charls_amd_) is additive: batch entry points working on DEVICE-resident frames, which is how the
 * engine is meant to be fed at scale (independent frames / scans are the sharding unit across wavefronts and GPUs).
 *
 * Plain C: opaque handles, plain pointers and sizes, no C++ or torch types cross this boundary
 * (the one exception, charls_get_jpegls_category, is inherited from the reference and documented there).
 */
#ifndef CHARLS_AMD_H
#define CHARLS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define CHARLS_AMD_API __attribute__((visibility("default")))
#else
#define CHARLS_AMD_API
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Types: include/charls/public_types.h:28-187 (enums), :934-1034 (structs; sizes 40/16/20/12 asserted at :1075-1078)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef int32_t charls_jpegls_errc; /* enum charls_jpegls_errc, int32-sized; values below */
enum
{
    CHARLS_JPEGLS_ERRC_SUCCESS = 0,
    CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY = 1,
    CHARLS_JPEGLS_ERRC_CALLBACK_FAILED = 2,
    CHARLS_JPEGLS_ERRC_DESTINATION_TOO_SMALL = 3,
    CHARLS_JPEGLS_ERRC_NEED_MORE_DATA = 4,
    CHARLS_JPEGLS_ERRC_INVALID_DATA = 5,
    CHARLS_JPEGLS_ERRC_ENCODING_NOT_SUPPORTED = 6,
    CHARLS_JPEGLS_ERRC_PARAMETER_VALUE_NOT_SUPPORTED = 7,
    CHARLS_JPEGLS_ERRC_COLOR_TRANSFORM_NOT_SUPPORTED = 8,
    CHARLS_JPEGLS_ERRC_JPEGLS_PRESET_EXTENDED_PARAMETER_TYPE_NOT_SUPPORTED = 9,
    CHARLS_JPEGLS_ERRC_JPEG_MARKER_START_BYTE_NOT_FOUND = 10,
    CHARLS_JPEGLS_ERRC_START_OF_IMAGE_MARKER_NOT_FOUND = 11,
    CHARLS_JPEGLS_ERRC_INVALID_SPIFF_HEADER = 12,
    CHARLS_JPEGLS_ERRC_UNKNOWN_JPEG_MARKER_FOUND = 13,
    CHARLS_JPEGLS_ERRC_UNEXPECTED_START_OF_SCAN_MARKER = 14,
    CHARLS_JPEGLS_ERRC_INVALID_MARKER_SEGMENT_SIZE = 15,
    CHARLS_JPEGLS_ERRC_DUPLICATE_START_OF_IMAGE_MARKER = 16,
    CHARLS_JPEGLS_ERRC_DUPLICATE_START_OF_FRAME_MARKER = 17,
    CHARLS_JPEGLS_ERRC_DUPLICATE_COMPONENT_ID_IN_SOF_SEGMENT = 18,
    CHARLS_JPEGLS_ERRC_UNEXPECTED_END_OF_IMAGE_MARKER = 19,
    CHARLS_JPEGLS_ERRC_INVALID_JPEGLS_PRESET_PARAMETER_TYPE = 20,
    CHARLS_JPEGLS_ERRC_MISSING_END_OF_SPIFF_DIRECTORY = 21,
    CHARLS_JPEGLS_ERRC_UNEXPECTED_RESTART_MARKER = 22,
    CHARLS_JPEGLS_ERRC_RESTART_MARKER_NOT_FOUND = 23,
    CHARLS_JPEGLS_ERRC_END_OF_IMAGE_MARKER_NOT_FOUND = 24,
    CHARLS_JPEGLS_ERRC_UNEXPECTED_DEFINE_NUMBER_OF_LINES_MARKER = 25,
    CHARLS_JPEGLS_ERRC_DEFINE_NUMBER_OF_LINES_MARKER_NOT_FOUND = 26,
    CHARLS_JPEGLS_ERRC_UNKNOWN_COMPONENT_ID = 27,
    CHARLS_JPEGLS_ERRC_ABBREVIATED_FORMAT_AND_SPIFF_HEADER_MISMATCH = 28,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_WIDTH = 29,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_HEIGHT = 30,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_BITS_PER_SAMPLE = 31,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COMPONENT_COUNT = 32,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_INTERLEAVE_MODE = 33,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_NEAR_LOSSLESS = 34,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_JPEGLS_PRESET_PARAMETERS = 35,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_COLOR_TRANSFORMATION = 36,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_MAPPING_TABLE_ID = 37,
    CHARLS_JPEGLS_ERRC_INVALID_PARAMETER_MAPPING_TABLE_CONTINUATION = 38,
    CHARLS_JPEGLS_ERRC_INVALID_OPERATION = 100,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT = 101,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_WIDTH = 102,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_HEIGHT = 103,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_BITS_PER_SAMPLE = 104,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COMPONENT_COUNT = 105,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_INTERLEAVE_MODE = 106,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_NEAR_LOSSLESS = 107,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_JPEGLS_PC_PARAMETERS = 108,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_COLOR_TRANSFORMATION = 109,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_SIZE = 110,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_STRIDE = 111,
    CHARLS_JPEGLS_ERRC_INVALID_ARGUMENT_ENCODING_OPTIONS = 112,
    /* additive, never produced by the reference: the engine has no CPU fallback, so a missing / failing GPU is an error */
    CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE = 200,
    CHARLS_AMD_ERRC_DEVICE_FAILURE = 201
};

/* The reference declares these as C enums (include/charls/public_types.h:90-187); an enum of these values is an int,
 * so 32-bit integer types with the same constant names give a C caller the same source and the same ABI. */
typedef int32_t charls_interleave_mode;
enum { CHARLS_INTERLEAVE_MODE_NONE = 0, CHARLS_INTERLEAVE_MODE_LINE = 1, CHARLS_INTERLEAVE_MODE_SAMPLE = 2 };
typedef int32_t charls_color_transformation;
enum { CHARLS_COLOR_TRANSFORMATION_NONE = 0, CHARLS_COLOR_TRANSFORMATION_HP1 = 1, CHARLS_COLOR_TRANSFORMATION_HP2 = 2,
       CHARLS_COLOR_TRANSFORMATION_HP3 = 3 };
typedef uint32_t charls_encoding_options;
enum { CHARLS_ENCODING_OPTIONS_NONE = 0, CHARLS_ENCODING_OPTIONS_EVEN_DESTINATION_SIZE = 1,
       CHARLS_ENCODING_OPTIONS_INCLUDE_VERSION_NUMBER = 2, CHARLS_ENCODING_OPTIONS_INCLUDE_PC_PARAMETERS_JAI = 4 };
typedef int32_t charls_compressed_data_format;
enum { CHARLS_COMPRESSED_DATA_FORMAT_UNKNOWN = 0, CHARLS_COMPRESSED_DATA_FORMAT_INTERCHANGE = 1,
       CHARLS_COMPRESSED_DATA_FORMAT_ABBREVIATED_IMAGE_DATA = 2, CHARLS_COMPRESSED_DATA_FORMAT_ABBREVIATED_TABLE_SPECIFICATION = 3 };
typedef int32_t charls_spiff_profile_id;
enum { CHARLS_SPIFF_PROFILE_ID_NONE = 0, CHARLS_SPIFF_PROFILE_ID_CONTINUOUS_TONE_BASE = 1,
       CHARLS_SPIFF_PROFILE_ID_CONTINUOUS_TONE_PROGRESSIVE = 2, CHARLS_SPIFF_PROFILE_ID_BI_LEVEL_FACSIMILE = 3,
       CHARLS_SPIFF_PROFILE_ID_CONTINUOUS_TONE_FACSIMILE = 4 };
typedef int32_t charls_spiff_color_space;
enum { CHARLS_SPIFF_COLOR_SPACE_BI_LEVEL_BLACK = 0, CHARLS_SPIFF_COLOR_SPACE_YCBCR_ITU_BT_709_VIDEO = 1,
       CHARLS_SPIFF_COLOR_SPACE_NONE = 2, CHARLS_SPIFF_COLOR_SPACE_YCBCR_ITU_BT_601_1_RGB = 3,
       CHARLS_SPIFF_COLOR_SPACE_YCBCR_ITU_BT_601_1_VIDEO = 4, CHARLS_SPIFF_COLOR_SPACE_GRAYSCALE = 8,
       CHARLS_SPIFF_COLOR_SPACE_PHOTO_YCC = 9, CHARLS_SPIFF_COLOR_SPACE_RGB = 10, CHARLS_SPIFF_COLOR_SPACE_CMY = 11,
       CHARLS_SPIFF_COLOR_SPACE_CMYK = 12, CHARLS_SPIFF_COLOR_SPACE_YCCK = 13, CHARLS_SPIFF_COLOR_SPACE_CIE_LAB = 14,
       CHARLS_SPIFF_COLOR_SPACE_BI_LEVEL_WHITE = 15 };
typedef int32_t charls_spiff_compression_type;
enum { CHARLS_SPIFF_COMPRESSION_TYPE_UNCOMPRESSED = 0, CHARLS_SPIFF_COMPRESSION_TYPE_MODIFIED_HUFFMAN = 1,
       CHARLS_SPIFF_COMPRESSION_TYPE_MODIFIED_READ = 2, CHARLS_SPIFF_COMPRESSION_TYPE_MODIFIED_MODIFIED_READ = 3,
       CHARLS_SPIFF_COMPRESSION_TYPE_JBIG = 4, CHARLS_SPIFF_COMPRESSION_TYPE_JPEG = 5, CHARLS_SPIFF_COMPRESSION_TYPE_JPEG_LS = 6 };
typedef int32_t charls_spiff_resolution_units;
enum { CHARLS_SPIFF_RESOLUTION_UNITS_ASPECT_RATIO = 0, CHARLS_SPIFF_RESOLUTION_UNITS_DOTS_PER_INCH = 1,
       CHARLS_SPIFF_RESOLUTION_UNITS_DOTS_PER_CENTIMETER = 2 };
enum { CHARLS_SPIFF_ENTRY_TAG_TRANSFER_CHARACTERISTICS = 2, CHARLS_SPIFF_ENTRY_TAG_COMPONENT_REGISTRATION = 3,
       CHARLS_SPIFF_ENTRY_TAG_IMAGE_ORIENTATION = 4, CHARLS_SPIFF_ENTRY_TAG_THUMBNAIL = 5, CHARLS_SPIFF_ENTRY_TAG_IMAGE_TITLE = 6,
       CHARLS_SPIFF_ENTRY_TAG_IMAGE_DESCRIPTION = 7, CHARLS_SPIFF_ENTRY_TAG_TIME_STAMP = 8,
       CHARLS_SPIFF_ENTRY_TAG_VERSION_IDENTIFIER = 9, CHARLS_SPIFF_ENTRY_TAG_CREATOR_IDENTIFICATION = 10,
       CHARLS_SPIFF_ENTRY_TAG_PROTECTION_INDICATOR = 11, CHARLS_SPIFF_ENTRY_TAG_COPYRIGHT_INFORMATION = 12,
       CHARLS_SPIFF_ENTRY_TAG_CONTACT_INFORMATION = 13, CHARLS_SPIFF_ENTRY_TAG_TILE_INDEX = 14,
       CHARLS_SPIFF_ENTRY_TAG_SCAN_INDEX = 15, CHARLS_SPIFF_ENTRY_TAG_SET_REFERENCE = 16 };
enum { CHARLS_MAPPING_TABLE_MISSING = -1 };

typedef struct charls_spiff_header
{
    charls_spiff_profile_id profile_id;
    int32_t component_count;
    uint32_t height;
    uint32_t width;
    charls_spiff_color_space color_space;
    int32_t bits_per_sample;
    charls_spiff_compression_type compression_type;
    charls_spiff_resolution_units resolution_units;
    uint32_t vertical_resolution;
    uint32_t horizontal_resolution;
} charls_spiff_header;

typedef struct charls_frame_info
{
    uint32_t width;
    uint32_t height;
    int32_t bits_per_sample;
    int32_t component_count;
} charls_frame_info;

typedef struct charls_jpegls_pc_parameters
{
    int32_t maximum_sample_value;
    int32_t threshold1;
    int32_t threshold2;
    int32_t threshold3;
    int32_t reset_value;
} charls_jpegls_pc_parameters;

typedef struct charls_mapping_table_info
{
    int32_t table_id;
    int32_t entry_size;
    uint32_t data_size;
} charls_mapping_table_info;

typedef int32_t (*charls_at_comment_handler)(const void* data, size_t size, void* user_context);
typedef int32_t (*charls_at_application_data_handler)(int32_t application_data_id, const void* data, size_t size,
                                                      void* user_context);

typedef struct charls_jpegls_encoder charls_jpegls_encoder;
typedef struct charls_jpegls_decoder charls_jpegls_decoder;

/* ------------------------------------------------------------------------------------------------------------------
 * Part 1a -- encoder: include/charls/charls_jpegls_encoder.h:25-318, implemented by src/charls_jpegls_encoder.cpp
 * ---------------------------------------------------------------------------------------------------------------- */
CHARLS_AMD_API charls_jpegls_encoder* charls_jpegls_encoder_create(void);                                   /* :25 */
CHARLS_AMD_API void charls_jpegls_encoder_destroy(const charls_jpegls_encoder* encoder);                    /* :33 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_frame_info(charls_jpegls_encoder* encoder,
                                                                       const charls_frame_info* frame_info); /* :42 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_near_lossless(charls_jpegls_encoder* encoder,
                                                                          int32_t near_lossless);            /* :52 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_encoding_options(charls_jpegls_encoder* encoder,
                                                                             charls_encoding_options options); /* :61 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_interleave_mode(charls_jpegls_encoder* encoder,
                                                                            charls_interleave_mode mode);    /* :72 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_preset_coding_parameters(
    charls_jpegls_encoder* encoder, const charls_jpegls_pc_parameters* preset_coding_parameters);           /* :85 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_color_transformation(
    charls_jpegls_encoder* encoder, charls_color_transformation color_transformation);                      /* :99 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_mapping_table_id(charls_jpegls_encoder* encoder,
                                                                             int32_t component_index,
                                                                             int32_t table_id);             /* :110 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_get_estimated_destination_size(
    const charls_jpegls_encoder* encoder, size_t* size_in_bytes);                                           /* :123 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_set_destination_buffer(charls_jpegls_encoder* encoder,
                                                                               void* destination_buffer,
                                                                               size_t destination_size_bytes); /* :136 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_standard_spiff_header(
    charls_jpegls_encoder* encoder, charls_spiff_color_space color_space, charls_spiff_resolution_units resolution_units,
    uint32_t vertical_resolution, uint32_t horizontal_resolution);                                          /* :152 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_spiff_header(charls_jpegls_encoder* encoder,
                                                                           const charls_spiff_header* spiff_header); /* :166 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_spiff_entry(charls_jpegls_encoder* encoder,
                                                                          uint32_t entry_tag, const void* entry_data,
                                                                          size_t entry_data_size_bytes);    /* :182 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_spiff_end_of_directory_entry(
    charls_jpegls_encoder* encoder);                                                                        /* :197 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_comment(charls_jpegls_encoder* encoder,
                                                                      const void* comment, size_t comment_size_bytes); /* :211 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_application_data(charls_jpegls_encoder* encoder,
                                                                               int32_t application_data_id,
                                                                               const void* application_data,
                                                                               size_t application_data_size_bytes); /* :228 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_write_mapping_table(charls_jpegls_encoder* encoder,
                                                                            int32_t table_id, int32_t entry_size,
                                                                            const void* table_data,
                                                                            size_t table_data_size_bytes);  /* :247 */
/* HOT PATH: :264-268 -> charls_jpegls_encoder::encode -> make_scan_codec<scan_encoder>()->encode_scan (src/...encoder.cpp:182-296) */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_encode_from_buffer(charls_jpegls_encoder* encoder,
                                                                           const void* source_buffer,
                                                                           size_t source_size_bytes, uint32_t stride);
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_encode_components_from_buffer(
    charls_jpegls_encoder* encoder, const void* source_buffer, size_t source_size_bytes, int32_t source_component_count,
    uint32_t stride);                                                                                       /* :284 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_create_abbreviated_format(charls_jpegls_encoder* encoder); /* :297 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_get_bytes_written(const charls_jpegls_encoder* encoder,
                                                                          size_t* bytes_written);           /* :306 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_encoder_rewind(charls_jpegls_encoder* encoder);             /* :316 */

/* ------------------------------------------------------------------------------------------------------------------
 * Part 1b -- decoder: include/charls/charls_jpegls_decoder.h:25-295, implemented by src/charls_jpegls_decoder.cpp
 * ---------------------------------------------------------------------------------------------------------------- */
CHARLS_AMD_API charls_jpegls_decoder* charls_jpegls_decoder_create(void);                                   /* :25 */
CHARLS_AMD_API void charls_jpegls_decoder_destroy(const charls_jpegls_decoder* decoder);                    /* :33 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_set_source_buffer(charls_jpegls_decoder* decoder,
                                                                          const void* source_buffer,
                                                                          size_t source_size_bytes);        /* :45 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_read_spiff_header(charls_jpegls_decoder* decoder,
                                                                          charls_spiff_header* spiff_header,
                                                                          int32_t* header_found);           /* :59 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_read_header(charls_jpegls_decoder* decoder);        /* :69 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_frame_info(const charls_jpegls_decoder* decoder,
                                                                       charls_frame_info* frame_info);      /* :81 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_near_lossless(const charls_jpegls_decoder* decoder,
                                                                          int32_t component_index,
                                                                          int32_t* near_lossless);          /* :95 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_interleave_mode(const charls_jpegls_decoder* decoder,
                                                                            int32_t component_index,
                                                                            charls_interleave_mode* interleave_mode); /* :109 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_preset_coding_parameters(
    const charls_jpegls_decoder* decoder, int32_t reserved, charls_jpegls_pc_parameters* preset_coding_parameters); /* :123 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_color_transformation(
    const charls_jpegls_decoder* decoder, charls_color_transformation* color_transformation);               /* :137 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_get_destination_size(const charls_jpegls_decoder* decoder,
                                                                             uint32_t stride,
                                                                             size_t* destination_size_bytes); /* :151 */
/* HOT PATH: :170-174 -> charls_jpegls_decoder::decode -> make_scan_codec<scan_decoder>()->decode_scan (src/...decoder.cpp:177-201) */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_decode_to_buffer(charls_jpegls_decoder* decoder,
                                                                         void* destination_buffer,
                                                                         size_t destination_size_bytes, uint32_t stride);
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_at_comment(charls_jpegls_decoder* decoder,
                                                                   charls_at_comment_handler handler,
                                                                   void* user_context);                     /* :186 */
CHARLS_AMD_API charls_jpegls_errc charls_jpegls_decoder_at_application_data(
    charls_jpegls_decoder* decoder, charls_at_application_data_handler handler, void* user_context);        /* :201 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_get_compressed_data_format(
    const charls_jpegls_decoder* decoder, charls_compressed_data_format* compressed_data_format);           /* :215 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_get_mapping_table_id(const charls_jpegls_decoder* decoder,
                                                                      int32_t component_index, int32_t* table_id); /* :229 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_find_mapping_table_index(const charls_jpegls_decoder* decoder,
                                                                          int32_t mapping_table_id, int32_t* index); /* :244 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_get_mapping_table_count(const charls_jpegls_decoder* decoder,
                                                                         int32_t* count);                   /* :257 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_get_mapping_table_info(const charls_jpegls_decoder* decoder,
                                                                        int32_t mapping_table_index,
                                                                        charls_mapping_table_info* mapping_table_info); /* :273 */
CHARLS_AMD_API charls_jpegls_errc charls_decoder_get_mapping_table_data(const charls_jpegls_decoder* decoder,
                                                                        int32_t mapping_table_index,
                                                                        void* mapping_table_data,
                                                                        size_t mapping_table_size_bytes);   /* :291 */

/* ------------------------------------------------------------------------------------------------------------------
 * Part 1c -- misc: include/charls/jpegls_error.h:12, jpegls_error.hpp:10, version.h:43-54, validate_spiff_header.h:23
 * ---------------------------------------------------------------------------------------------------------------- */
CHARLS_AMD_API const char* charls_get_error_message(charls_jpegls_errc error_value);
CHARLS_AMD_API const void* charls_get_jpegls_category(void); /* really `const std::error_category*`, as in the reference */
CHARLS_AMD_API const char* charls_get_version_string(void);
CHARLS_AMD_API void charls_get_version_number(int32_t* major, int32_t* minor, int32_t* patch);
CHARLS_AMD_API charls_jpegls_errc charls_validate_spiff_header(const charls_spiff_header* spiff_header,
                                                               const charls_frame_info* frame_info);

/* ------------------------------------------------------------------------------------------------------------------
 * Part 2 -- additive batch API on DEVICE memory (no reference counterpart; never changes part 1 semantics).
 *
 * A batch is `frame_count` independent frames with identical coding parameters.  Frame f's pixels start at
 * d_frames + f * frame_pitch_bytes in the reference's user layout (planar for ILV_NONE, pixel-interleaved otherwise,
 * `stride` bytes between rows, 0 = minimal).  Frame f's complete .jls file is produced at / read from
 * d_streams + f * stream_pitch_bytes.  `sizes` and `errcs` are HOST arrays of frame_count elements.
 * `hip_stream` is a hipStream_t (NULL = default stream); the calls return after the stream work has completed.
 * Every frame gets the bytes and the errc the part-1 encoder/decoder would give it with a destination buffer of
 * stream_pitch_bytes (encode) or a source buffer of sizes[f] bytes (decode).
 * Slots need no alignment, but the decoders load whole 16-byte aligned groups: the device allocation that holds
 * d_streams must be readable from the 16-byte boundary at or before its first slot to the one at or after the end of its
 * last slot (any hipMalloc'ed buffer is; a slot that ends on the last byte of a sub-allocated pool may not be).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct charls_amd_codec_params
{
    charls_frame_info frame_info;
    int32_t near_lossless;
    charls_interleave_mode interleave_mode;
    charls_color_transformation color_transformation;
    charls_jpegls_pc_parameters preset_coding_parameters; /* all zero = defaults */
    charls_encoding_options encoding_options;
    uint32_t restart_interval; /* lines per restart interval, 0 = none (encode: extension below; decode: from DRI) */
} charls_amd_codec_params;

CHARLS_AMD_API charls_jpegls_errc charls_amd_encode_batch_device(const charls_amd_codec_params* params,
                                                                 uint32_t frame_count, const void* d_frames,
                                                                 size_t frame_pitch_bytes, uint32_t stride,
                                                                 void* d_streams, size_t stream_pitch_bytes,
                                                                 uint64_t* sizes, charls_jpegls_errc* errcs,
                                                                 void* hip_stream);

CHARLS_AMD_API charls_jpegls_errc charls_amd_decode_batch_device(uint32_t frame_count, const void* d_streams,
                                                                 size_t stream_pitch_bytes, const uint64_t* sizes,
                                                                 void* d_frames, size_t frame_pitch_bytes,
                                                                 uint32_t stride, charls_amd_codec_params* params_out,
                                                                 charls_jpegls_errc* errcs, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Part 2b -- several GPUs from one process (frames are the sharding unit; no exchange while coding; the only
 * collective is the hand-over of the finished bitstreams).  The frames of a batch are dealt to shards; shard s lives on
 * device shards[s].device with its frames and its stream slots in that device's memory, at the pitches given to the call.
 * A worker thread per shard, bound to the shard's device, runs charls_amd_encode_batch_device / _decode_batch_device on the
 * shard (the threads belong to a context that lives across calls, see charls_amd_devices below); sizes / errcs are HOST arrays over all frames in shard order (shard 0's frames first).  Every frame gets exactly
 * the bytes and the errc of part 1, whatever the number of shards.
 *
 * encode, gather != NULL: after coding, every shard's streams are brought together, back to back and in frame order, in
 * gather->d_gathered on the device of shard gather->root_shard; offsets[f] is where frame f starts (frames that failed
 * take no room), *total_bytes the end.  The sizes are exchanged first (a prefix sum over host values here), then every
 * other shard sends each stream -- exactly sizes[f] bytes -- to its place: with RCCL (ncclSend / ncclRecv pairs in groups,
 * point to point over xGMI; librccl.so is opened at run time) or with peer copies (hipMemcpyPeerAsync).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct charls_amd_device_shard
{
    int32_t device;       /* HIP device ordinal */
    uint32_t frame_count; /* frames of this shard (may be 0) */
    const void* d_frames; /* encode: source frames; decode: destination frames (written) */
    void* d_streams;      /* encode: destination slots; decode: source slots */
    void* hip_stream;     /* a hipStream_t of that device, or NULL */
} charls_amd_device_shard;

typedef enum charls_amd_transport
{
    CHARLS_AMD_TRANSPORT_AUTO = 0,       /* RCCL when it can be loaded and the shards sit on distinct devices, else peer copies */
    CHARLS_AMD_TRANSPORT_RCCL = 1,       /* fail when RCCL is not usable */
    CHARLS_AMD_TRANSPORT_PEER_COPIES = 2
} charls_amd_transport;

typedef struct charls_amd_gather
{
    uint32_t root_shard;
    void* d_gathered;      /* on the root shard's device */
    size_t capacity_bytes;
    uint64_t* offsets;     /* HOST, one per frame */
    uint64_t* total_bytes; /* HOST, may be NULL */
    charls_amd_transport transport;
} charls_amd_gather;

CHARLS_AMD_API charls_jpegls_errc charls_amd_encode_batch_devices(const charls_amd_codec_params* params,
                                                                  uint32_t shard_count, const charls_amd_device_shard* shards,
                                                                  size_t frame_pitch_bytes, uint32_t stride,
                                                                  size_t stream_pitch_bytes, uint64_t* sizes,
                                                                  charls_jpegls_errc* errcs, const charls_amd_gather* gather);

CHARLS_AMD_API charls_jpegls_errc charls_amd_decode_batch_devices(uint32_t shard_count, const charls_amd_device_shard* shards,
                                                                  size_t stream_pitch_bytes, const uint64_t* sizes,
                                                                  size_t frame_pitch_bytes, uint32_t stride,
                                                                  charls_amd_codec_params* params_out, charls_jpegls_errc* errcs);

/* The context behind the two calls above.  It lives across calls and owns a worker thread per (device, shard ordinal on
 * that device), bound to its device for life -- the encoder's work areas belong to the thread that made them, so a second
 * call of the same shape allocates nothing -- and the RCCL communicator and exchange streams of the last gathered call
 * (re-made only when the list of devices changes).  One call at a time per context; different contexts are independent.
 * charls_amd_encode_batch_devices / charls_amd_decode_batch_devices run on a process-wide default context, which is also
 * what a NULL `context` argument means below; charls_amd_devices_destroy(NULL) releases everything the default context
 * holds (its threads end, their work areas are freed, the communicator is destroyed). */
typedef struct charls_amd_devices charls_amd_devices;
CHARLS_AMD_API charls_amd_devices* charls_amd_devices_create(void); /* NULL when out of memory */
CHARLS_AMD_API void charls_amd_devices_destroy(charls_amd_devices* context);
CHARLS_AMD_API charls_jpegls_errc charls_amd_devices_encode_batch(charls_amd_devices* context, const charls_amd_codec_params* params,
                                                                  uint32_t shard_count, const charls_amd_device_shard* shards,
                                                                  size_t frame_pitch_bytes, uint32_t stride,
                                                                  size_t stream_pitch_bytes, uint64_t* sizes,
                                                                  charls_jpegls_errc* errcs, const charls_amd_gather* gather);
CHARLS_AMD_API charls_jpegls_errc charls_amd_devices_decode_batch(charls_amd_devices* context, uint32_t shard_count,
                                                                  const charls_amd_device_shard* shards, size_t stream_pitch_bytes,
                                                                  const uint64_t* sizes, size_t frame_pitch_bytes, uint32_t stride,
                                                                  charls_amd_codec_params* params_out, charls_jpegls_errc* errcs);
CHARLS_AMD_API uint64_t charls_amd_devices_work_area_bytes(charls_amd_devices* context); /* HBM held by the context's workers */
CHARLS_AMD_API charls_jpegls_errc charls_amd_devices_release_work_areas(charls_amd_devices* context); /* (the threads stay) */

/* Extension: restart intervals on the encoder of part 1.  `lines` rows per interval (0 = none, the default) are coded
 * independently -- by different wavefronts at the same time -- and separated by RSTm markers; a DRI segment announces
 * the interval.  The reference's encoder has no equivalent (its output never contains restart markers); its decoder
 * reads these streams (reference src/jpeg_stream_reader.cpp:586-607, src/scan_decoder.hpp:335-349), and this library's
 * decoder decodes the intervals in parallel.  Must be called before the first encode_* call of an image. */
CHARLS_AMD_API charls_jpegls_errc charls_amd_jpegls_encoder_set_restart_interval(charls_jpegls_encoder* encoder,
                                                                                 uint32_t lines);

/* Engine selection for the lossless single-component encoder: 0 = automatic, 1 = force the one-wavefront-per-scan
 * kernel, 2 = force the parallel pipeline (returns invalid_argument when the scan is not eligible). Process-wide. */
CHARLS_AMD_API charls_jpegls_errc charls_amd_set_encode_engine(int32_t engine);

/* HBM kept by the library for its own work areas (the lossless encoder's per-scan work area of 8 B per sample of up to 8
 * bits -- 10 B per wider sample -- plus the unstuffed stream, the private buffers of restart intervals).  The batch entry
 * points of part 2 keep them per calling thread and device (a thread that moves to another device gets new ones there);
 * they grow on demand and stay allocated between calls.  The limit is process-wide: 0 (the default) = a quarter of the
 * device's memory, and never more than what is free minus 8 GiB.  A batch larger than the limit allows is coded in several
 * passes; when not even one work area can be allocated the encoder falls back to its one-wavefront-per-scan kernel, which
 * needs none (counted: charls_amd_engine_counters [4]).
 *
 * The host-pointer encoder / decoder of part 1 (THREADING, as the reference: distinct handles are independent, callers
 * scale by threads x handles): calls that arrive together are merged into one kernel launch (charls_amd_engine_counters),
 * and the merged encoder launches of ALL threads run on ONE set of work areas per device -- a pool of 256 threads holds
 * one arena, not 256.  That set never grows beyond the limit nor beyond an eighth of the device's memory (larger batches
 * take more passes) and stays allocated between calls that follow each other: giving gigabytes back to the driver and
 * asking for them again costs seconds, and hipFree waits for every kernel on the device.  Handles share a process-wide pool
 * of device buffers, streams and pinned staging areas (idle sets of at most 512 MiB each, per device at most 18 GiB or the
 * limit, whichever is less: callers create a handle per image, creating these per handle costs more than coding a frame).
 * None of it stays for long: once no coding call of part 1 has run for two seconds (CHARLS_AMD_IDLE_RELEASE_MS; 0 = keep) the
 * shared set, the idle pool and every block whose hipFree had been put off are given back by a housekeeping thread of the
 * library (charls_amd_engine_counters [6] - [8]).  charls_amd_release_work_areas frees the calling thread's areas, the shared
 * set and the idle pool at once. */
CHARLS_AMD_API charls_jpegls_errc charls_amd_set_workspace_limit(uint64_t bytes);
CHARLS_AMD_API charls_jpegls_errc charls_amd_release_work_areas(void);
CHARLS_AMD_API uint64_t charls_amd_work_area_bytes(void); /* the calling thread's + the shared set of part 1, currently allocated */

/* What the engine did with the calls of part 1 since the library was loaded (process-wide): out[0] scan submissions of
 * the host-pointer ABI, out[1] kernel launches they took, out[2] submissions that shared their launch with another call,
 * out[3] scans of the largest launch, out[4] scans the lossless pipeline was eligible for that were coded by the
 * one-wavefront kernel because no work area could be allocated, out[5] merged launches that ran out of memory and whose
 * calls were then run one by one, out[6] bytes of device memory in the idle pool of handle resources, out[7] bytes whose
 * hipFree has been put off (a decoder launch is running), out[8] how often the housekeeping thread gave memory back,
 * out[9] scans a speed-path decoder handed to the exact decoder (streams that are damaged or that end unusually; a valid
 * stream that is counted here lost its speed path).  Returns the number of values written (10 at most). */
CHARLS_AMD_API int32_t charls_amd_engine_counters(uint64_t* out, int32_t capacity);

/* Test and measurement knobs (charls_amd/csrc/device/knobs.h has the list: DECODE_GROUP, JOB_EVENTS, TILE_SAMPLES,
 * COALESCE, ...).  The environment (CHARLS_AMD_<NAME>) is read ONCE, when the first knob is looked at; after that only this
 * call changes a value.  `name` with or without the CHARLS_AMD_ prefix; `value` INT64_MIN clears the knob (the engine's own
 * rule applies again).  Process-wide, not synchronised with calls in flight.  invalid_argument for an unknown name. */
CHARLS_AMD_API charls_jpegls_errc charls_amd_debug_set_knob(const char* name, int64_t value);

/* Milliseconds of GPU time (hipEvent) the last batch call on this thread spent in its kernels, by stage:
 * out[0] total, out[1] dominant kernel, out[2..7] stage breakdown. Returns the number of values. */
CHARLS_AMD_API int32_t charls_amd_last_timings(double* out, int32_t capacity);

/* The lossless encoder codes the chain of every context in JOBS that start from a guessed state and are checked against
 * their predecessors afterwards; a job whose guess was wrong is coded again from the true state, so the bytes
 * never depend on the guesses -- only the time does.  Process-wide totals since the library was loaded:
 * out[0] jobs of the regular chains, out[1] how many of them were coded again, out[2] / out[3] the same for the run
 * chain; out[4] segments of the event lists of the rarer run-interruption context that were walked from a guessed state,
 * out[5] scans whose list was walked again serially because such a guess was wrong.  Frames whose jobs are mostly coded
 * again (full-range noise) encode at the speed of one lane per chain: a caller can tell from these counters.  Returns the
 * number of values written (6 at most). */
CHARLS_AMD_API int32_t charls_amd_speculation_counters(uint64_t* out, int32_t capacity);

/* 0 when a gfx950 device is usable, otherwise CHARLS_AMD_ERRC_DEVICE_UNAVAILABLE. */
CHARLS_AMD_API charls_jpegls_errc charls_amd_device_status(void);

#ifdef __cplusplus
}
#endif
#endif
