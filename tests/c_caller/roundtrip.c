/* A plain C99 caller of the CharLS C API, the way an application built against CharLS uses it (the reference's own
 * example of such a caller is samples/convert-c/main.c:192-302: create, set_frame_info, set_interleave_mode,
 * set_near_lossless, get_estimated_destination_size, set_destination_buffer, write_standard_spiff_header,
 * encode_from_buffer, get_bytes_written; then the decoder calls).  TEST ONLY.
 *
 * Built twice by __graft_entry__.build():
 *   - against this repository's header (include/charls_amd.h), and
 *   - where the reference tree is present, against the REFERENCE's own headers (<charls/charls.h>), unchanged,
 * and both times linked with -lcharls, i.e. against the SONAME libcharls.so.3 that charls_amd/lib provides: the link
 * test behind INTEGRATION.md's "a caller built for CharLS runs on this library".  Exit code 0 = round trips are exact;
 * 3 = the library reported an error (printed; 200 = no usable GPU, the library has no CPU codec).
 */
#ifdef USE_REFERENCE_HEADERS
#include <charls/charls.h>
#else
#include "charls_amd.h"
#endif

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int fail(const char* what, charls_jpegls_errc error)
{
    printf("c_caller: %s failed: errc %d (%s)\n", what, (int)error, charls_get_error_message(error));
    return 3;
}

/* width x height RGB picture, pixel interleaved: smooth ramps with a few flat blocks (run mode) */
static uint8_t* make_picture(uint32_t width, uint32_t height)
{
    uint8_t* p = (uint8_t*)malloc((size_t)width * height * 3);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
        {
            uint8_t* px = p + ((size_t)y * width + x) * 3;
            const int flat = ((x >> 4) + (y >> 3)) % 5 == 0;
            px[0] = (uint8_t)(flat ? 200 : (x * 2 + y) & 0xFF);
            px[1] = (uint8_t)(flat ? 100 : (x + y * 3) & 0xFF);
            px[2] = (uint8_t)(flat ? 50 : ((x ^ y) * 5) & 0xFF);
        }
    return p;
}

static int round_trip(const uint8_t* pixels, uint32_t width, uint32_t height, charls_interleave_mode mode, int near_lossless)
{
    const size_t pixel_bytes = (size_t)width * height * 3;
    uint8_t* source = (uint8_t*)malloc(pixel_bytes);
    if (mode == CHARLS_INTERLEAVE_MODE_NONE)
    { /* planar layout: three planes */
        for (size_t i = 0; i < (size_t)width * height; ++i)
            for (int c = 0; c < 3; ++c)
                source[(size_t)c * width * height + i] = pixels[i * 3 + c];
    }
    else
        memcpy(source, pixels, pixel_bytes);

    charls_jpegls_encoder* encoder = charls_jpegls_encoder_create();
    if (!encoder)
        return fail("encoder_create", CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY);
    charls_frame_info info;
    info.width = width;
    info.height = height;
    info.bits_per_sample = 8;
    info.component_count = 3;
    charls_jpegls_errc e = charls_jpegls_encoder_set_frame_info(encoder, &info);
    if (!e)
        e = charls_jpegls_encoder_set_interleave_mode(encoder, mode);
    if (!e)
        e = charls_jpegls_encoder_set_near_lossless(encoder, near_lossless);
    size_t capacity = 0;
    if (!e)
        e = charls_jpegls_encoder_get_estimated_destination_size(encoder, &capacity);
    if (e)
        return fail("encoder set-up", e);
    uint8_t* encoded = (uint8_t*)malloc(capacity);
    e = charls_jpegls_encoder_set_destination_buffer(encoder, encoded, capacity);
    if (!e)
        e = charls_jpegls_encoder_write_standard_spiff_header(encoder, CHARLS_SPIFF_COLOR_SPACE_RGB,
                                                              CHARLS_SPIFF_RESOLUTION_UNITS_ASPECT_RATIO, 1, 1);
    if (!e)
        e = charls_jpegls_encoder_encode_from_buffer(encoder, source, pixel_bytes, 0);
    size_t encoded_size = 0;
    if (!e)
        e = charls_jpegls_encoder_get_bytes_written(encoder, &encoded_size);
    charls_jpegls_encoder_destroy(encoder);
    if (e)
        return fail("encode", e);

    charls_jpegls_decoder* decoder = charls_jpegls_decoder_create();
    if (!decoder)
        return fail("decoder_create", CHARLS_JPEGLS_ERRC_NOT_ENOUGH_MEMORY);
    e = charls_jpegls_decoder_set_source_buffer(decoder, encoded, encoded_size);
    charls_spiff_header spiff;
    int32_t spiff_found = 0;
    if (!e)
        e = charls_jpegls_decoder_read_spiff_header(decoder, &spiff, &spiff_found);
    if (!e)
        e = charls_jpegls_decoder_read_header(decoder);
    charls_frame_info read_back;
    if (!e)
        e = charls_jpegls_decoder_get_frame_info(decoder, &read_back);
    charls_interleave_mode read_mode = CHARLS_INTERLEAVE_MODE_NONE;
    if (!e)
        e = charls_jpegls_decoder_get_interleave_mode(decoder, 0, &read_mode);
    size_t decoded_size = 0;
    if (!e)
        e = charls_jpegls_decoder_get_destination_size(decoder, 0, &decoded_size);
    if (e)
        return fail("decoder set-up", e);
    if (!spiff_found || spiff.width != width || spiff.height != height || read_back.width != width || read_back.height != height ||
        read_back.component_count != 3 || read_mode != mode || decoded_size != pixel_bytes)
    {
        printf("c_caller: header read back differs\n");
        return 3;
    }
    uint8_t* decoded = (uint8_t*)malloc(decoded_size);
    e = charls_jpegls_decoder_decode_to_buffer(decoder, decoded, decoded_size, 0);
    charls_jpegls_decoder_destroy(decoder);
    if (e)
        return fail("decode", e);
    int worst = 0;
    for (size_t i = 0; i < pixel_bytes; ++i)
    {
        const int d = abs((int)decoded[i] - (int)source[i]);
        worst = d > worst ? d : worst;
    }
    printf("c_caller: mode %d near %d: %zu -> %zu bytes, max |x - x'| = %d\n", (int)mode, near_lossless, pixel_bytes, encoded_size,
           worst);
    free(decoded);
    free(encoded);
    free(source);
    return worst <= near_lossless ? 0 : 3;
}

int main(void)
{
    const uint32_t width = 200, height = 64;
    uint8_t* pixels = make_picture(width, height);
    printf("c_caller: library %s\n", charls_get_version_string());
    int rc = round_trip(pixels, width, height, CHARLS_INTERLEAVE_MODE_SAMPLE, 0);
    if (!rc)
        rc = round_trip(pixels, width, height, CHARLS_INTERLEAVE_MODE_NONE, 0);
    if (!rc)
        rc = round_trip(pixels, width, height, CHARLS_INTERLEAVE_MODE_LINE, 2);
    free(pixels);
    if (!rc)
        printf("c_caller ok\n");
    return rc;
}
