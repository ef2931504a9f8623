/*
 * jls_oracle.c -- scalar C99 restatement of the CharLS JPEG-LS codec (TEST INFRASTRUCTURE ONLY).
 *
 * See jls_oracle.h for the role of this file. Every function cites the reference file:line it restates
 * (paths relative to /root/reference). Nothing here is used by the product path.
 *
 * Known, documented deviation: when the source buffer ends inside entropy data WITHOUT any marker and the bit
 * reader has a negative valid-bit balance, the reference keeps reading zero bits (src/scan_decoder.hpp:259-268,
 * practically an endless loop); this restatement reports invalid_data instead.
 */
#include "jls_oracle.h"

#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

enum
{
    E_OK = 0,
    E_NOT_ENOUGH_MEMORY = 1,
    E_DESTINATION_TOO_SMALL = 3,
    E_NEED_MORE_DATA = 4,
    E_INVALID_DATA = 5,
    E_ENCODING_NOT_SUPPORTED = 6,
    E_PARAMETER_VALUE_NOT_SUPPORTED = 7,
    E_COLOR_TRANSFORM_NOT_SUPPORTED = 8,
    E_PRESET_EXTENDED_NOT_SUPPORTED = 9,
    E_MARKER_START_BYTE_NOT_FOUND = 10,
    E_SOI_NOT_FOUND = 11,
    E_UNKNOWN_MARKER = 13,
    E_UNEXPECTED_SOS = 14,
    E_INVALID_SEGMENT_SIZE = 15,
    E_DUPLICATE_SOI = 16,
    E_DUPLICATE_SOF = 17,
    E_DUPLICATE_COMPONENT_ID = 18,
    E_UNEXPECTED_EOI = 19,
    E_INVALID_PRESET_TYPE = 20,
    E_UNEXPECTED_RESTART_MARKER = 22,
    E_RESTART_MARKER_NOT_FOUND = 23,
    E_EOI_NOT_FOUND = 24,
    E_INVALID_PARAMETER_WIDTH = 29,
    E_INVALID_PARAMETER_HEIGHT = 30,
    E_INVALID_PARAMETER_BPS = 31,
    E_INVALID_PARAMETER_COMPONENT_COUNT = 32,
    E_INVALID_PARAMETER_ILV = 33,
    E_INVALID_PARAMETER_NEAR = 34,
    E_INVALID_PARAMETER_PC = 35,
    E_INVALID_PARAMETER_COLOR_TRANSFORMATION = 36,
    E_INVALID_OPERATION = 100,
    E_INVALID_ARGUMENT = 101,
    E_INVALID_ARGUMENT_WIDTH = 102,
    E_INVALID_ARGUMENT_HEIGHT = 103,
    E_INVALID_ARGUMENT_BPS = 104,
    E_INVALID_ARGUMENT_COMPONENT_COUNT = 105,
    E_INVALID_ARGUMENT_ILV = 106,
    E_INVALID_ARGUMENT_NEAR = 107,
    E_INVALID_ARGUMENT_PC = 108,
    E_INVALID_ARGUMENT_COLOR_TRANSFORMATION = 109,
    E_INVALID_ARGUMENT_SIZE = 110,
    E_INVALID_ARGUMENT_STRIDE = 111
};

/* src/scan_codec.hpp:18-19 */
static const int J[32] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 9, 10, 11, 12, 13, 14, 15};

typedef struct
{
    int32_t a, b, c, n;
} reg_ctx; /* src/regular_mode_context.hpp:139-143 */

typedef struct
{
    int32_t ritype, a, n, nn;
} run_ctx; /* src/run_mode_context.hpp:117-122 */

typedef struct
{
    jmp_buf fail;
    /* scan description */
    uint32_t w, h;
    int nc; /* components in this scan */
    int ilv, near, bpp, xform;
    int maxval, range, qbpp, limit, t1, t2, t3, reset;
    uint32_t restart_interval;
    /* model state (src/scan_codec.hpp:199-213) */
    reg_ctx reg[365];
    run_ctx rc[2];
    int run_index;
    int8_t* qstore;
    const int8_t* q;
    uint16_t* lines; /* 2 x nc x (w + 2) */
    /* bit writer (src/scan_encoder.hpp:192-202) */
    uint8_t* wpos;
    size_t wremaining, wwritten;
    uint32_t wbuf;
    int wfree, wff;
    /* bit reader (src/scan_decoder.hpp:354-360) */
    const uint8_t* rpos;
    const uint8_t* rend;
    const uint8_t* rff;
    uint64_t rcache;
    int rvalid;
    uint32_t rst_counter;
} codec;

static void fail(codec* c, int errc)
{
    longjmp(c->fail, errc);
}

/* ---------------------------------------------------------------- arithmetic primitives */

static int log2_ceiling(int32_t n) /* src/jpegls_algorithm.hpp:13-25 */
{
    int x = 0;
    while (n > (1 << x))
        ++x;
    return x;
}

int32_t jls_oracle_map_error(int32_t e) /* src/jpegls_algorithm.hpp:67-73 */
{
    return e >= 0 ? 2 * e : -2 * e - 1;
}

int32_t jls_oracle_unmap_error(int32_t m) /* src/jpegls_algorithm.hpp:80-86 */
{
    return (m & 1) ? -(m >> 1) - 1 : (m >> 1);
}

static int32_t sgn(int32_t n) /* src/jpegls_algorithm.hpp:89-93: +1 for n >= 0 */
{
    return n < 0 ? -1 : 1;
}

int32_t jls_oracle_med(int32_t ra, int32_t rb, int32_t rc) /* src/jpegls_algorithm.hpp:143-161 */
{
    const int32_t lo = ra < rb ? ra : rb;
    const int32_t hi = ra < rb ? rb : ra;
    if (rc >= hi)
        return lo;
    if (rc <= lo)
        return hi;
    return ra + rb - rc;
}

int jls_oracle_quantize_gradient(int32_t di, int32_t t1, int32_t t2, int32_t t3, int32_t near)
{ /* src/jpegls_algorithm.hpp:173-194 */
    if (di <= -t3)
        return -4;
    if (di <= -t2)
        return -3;
    if (di <= -t1)
        return -2;
    if (di < -near)
        return -1;
    if (di <= near)
        return 0;
    if (di < t1)
        return 1;
    if (di < t2)
        return 2;
    if (di < t3)
        return 3;
    return 4;
}

static int32_t clamp_pc(int32_t i, int32_t j, int32_t maxval) /* src/jpegls_preset_coding_parameters.hpp:16-20 */
{
    return (i > maxval || i < j) ? j : i;
}

static int32_t imax(int32_t a, int32_t b)
{
    return a > b ? a : b;
}
static int32_t imin(int32_t a, int32_t b)
{
    return a < b ? a : b;
}

void jls_oracle_default_pc(int32_t maxval, int32_t near, int32_t out[5]) /* src/jpegls_preset_coding_parameters.hpp:24-57 */
{
    int32_t t1, t2, t3;
    if (maxval >= 128)
    {
        const int32_t f = (imin(maxval, 4095) + 128) / 256;
        t1 = clamp_pc(f * (3 - 2) + 2 + 3 * near, near + 1, maxval);
        t2 = clamp_pc(f * (7 - 3) + 3 + 5 * near, t1, maxval);
        t3 = clamp_pc(f * (21 - 4) + 4 + 7 * near, t2, maxval);
    }
    else
    {
        const int32_t f = 256 / (maxval + 1);
        t1 = clamp_pc(imax(2, 3 / f + 3 * near), near + 1, maxval);
        t2 = clamp_pc(imax(3, 7 / f + 5 * near), t1, maxval);
        t3 = clamp_pc(imax(4, 21 / f + 7 * near), t2, maxval);
    }
    out[0] = maxval;
    out[1] = t1;
    out[2] = t2;
    out[3] = t3;
    out[4] = 64;
}

/* src/jpegls_preset_coding_parameters.hpp:61-130: is_default / is_valid */
static int pc_is_default(const int32_t p[5], const int32_t d[5])
{
    if (!p[0] && !p[1] && !p[2] && !p[3] && !p[4])
        return 1;
    return p[0] == d[0] && p[1] == d[1] && p[2] == d[2] && p[3] == d[3] && p[4] == d[4];
}

static int pc_validate(const int32_t p[5], int32_t bit_maxval, int32_t near, int32_t out[5])
{
    int32_t d[5];
    if (p[0] != 0 && (p[0] < 1 || p[0] > bit_maxval))
        return 0;
    const int32_t maxval = p[0] != 0 ? p[0] : bit_maxval;
    if (p[1] != 0 && (p[1] < near + 1 || p[1] > maxval))
        return 0;
    jls_oracle_default_pc(maxval, near, d);
    const int32_t t1 = p[1] != 0 ? p[1] : d[1];
    if (p[2] != 0 && (p[2] < t1 || p[2] > maxval))
        return 0;
    const int32_t t2 = p[2] != 0 ? p[2] : d[2];
    if (p[3] != 0 && (p[3] < t2 || p[3] > maxval))
        return 0;
    if (p[4] != 0 && (p[4] < 3 || p[4] > imax(255, maxval)))
        return 0;
    out[0] = maxval;
    out[1] = t1;
    out[2] = t2;
    out[3] = p[3] != 0 ? p[3] : d[3];
    out[4] = p[4] != 0 ? p[4] : d[4];
    return 1;
}

/* ---------------------------------------------------------------- traits (src/default_traits.hpp) */

static int32_t correct_prediction(const codec* c, int32_t p) /* src/default_traits.hpp:110-116 */
{
    if (p < 0)
        return 0;
    if (p > c->maxval)
        return c->maxval;
    return p;
}

static int32_t compute_error_value(const codec* c, int32_t e) /* src/default_traits.hpp:77-80,123-139,157-163 */
{
    const int32_t d = 2 * c->near + 1;
    e = e > 0 ? (e + c->near) / d : -(c->near - e) / d;
    if (e < 0)
        e += c->range;
    if (e >= (c->range + 1) / 2)
        e -= c->range;
    return e;
}

static int32_t reconstruct(const codec* c, int32_t predicted, int32_t e) /* src/default_traits.hpp:83-87,172-184 */
{
    int32_t v = predicted + e * (2 * c->near + 1);
    if (v < -c->near)
        v += c->range * (2 * c->near + 1);
    else if (v > c->maxval + c->near)
        v -= c->range * (2 * c->near + 1);
    return correct_prediction(c, v);
}

static int is_near(const codec* c, int32_t a, int32_t b) /* src/default_traits.hpp:90-93 */
{
    const int32_t d = a - b;
    return (d < 0 ? -d : d) <= c->near;
}

/* ---------------------------------------------------------------- context model */

static void init_model(codec* c) /* src/scan_codec.hpp:160-171, src/jpegls_algorithm.hpp:56-60 */
{
    const int32_t a0 = imax(2, (c->range + 32) / 64);
    for (int i = 0; i < 365; ++i)
    {
        c->reg[i].a = a0;
        c->reg[i].b = 0;
        c->reg[i].c = 0;
        c->reg[i].n = 1;
    }
    for (int i = 0; i < 2; ++i)
    {
        c->rc[i].ritype = i;
        c->rc[i].a = a0;
        c->rc[i].n = 1;
        c->rc[i].nn = 0;
    }
    c->run_index = 0;
}

static int reg_k(codec* c, const reg_ctx* x) /* src/regular_mode_context.hpp:99-111 (k >= 16 -> invalid_data) */
{
    int k = 0;
    while (k < 16 && (x->n << k) < x->a)
        ++k;
    if (k == 16)
        fail(c, E_INVALID_DATA);
    return k;
}

static void reg_update(codec* c, reg_ctx* x, int32_t e) /* src/regular_mode_context.hpp:45-93 */
{
    x->a += e < 0 ? -e : e;
    x->b += e * (2 * c->near + 1);
    if (x->a >= 65536 * 256 || x->b >= 65536 * 256 || x->b <= -65536 * 256)
        fail(c, E_INVALID_DATA);
    if (x->n == c->reset)
    {
        x->a >>= 1;
        x->b >>= 1; /* arithmetic shift of a possibly negative value, as the reference relies on */
        x->n >>= 1;
    }
    ++x->n;
    if (x->b + x->n <= 0)
    {
        x->b += x->n;
        if (x->b <= -x->n)
            x->b = -x->n + 1;
        if (x->c > -128)
            --x->c;
    }
    else if (x->b > 0)
    {
        x->b -= x->n;
        if (x->b > 0)
            x->b = 0;
        if (x->c < 127)
            ++x->c;
    }
}

static int run_k(codec* c, const run_ctx* x, int checked) /* src/run_mode_context.hpp:34-62 */
{
    const int64_t temp = (int64_t)x->a + (int64_t)(x->n >> 1) * x->ritype;
    int64_t n_test = x->n;
    int k = 0;
    for (; n_test < temp; ++k)
    {
        n_test <<= 1;
        if (k > 32)
        {
            if (checked)
                fail(c, E_INVALID_DATA);
            break;
        }
    }
    return k;
}

static int run_map(const run_ctx* x, int32_t e, int k) /* src/run_mode_context.hpp:103-115 */
{
    if (k == 0 && e > 0 && 2 * x->nn < x->n)
        return 1;
    if (e < 0 && 2 * x->nn >= x->n)
        return 1;
    if (e < 0 && k != 0)
        return 1;
    return 0;
}

static void run_update(codec* c, run_ctx* x, int32_t e, int32_t em) /* src/run_mode_context.hpp:65-83 */
{
    if (e < 0)
        ++x->nn;
    x->a += (em + 1 - x->ritype) >> 1;
    if (x->n == c->reset)
    {
        x->a >>= 1;
        x->n >>= 1;
        x->nn >>= 1;
    }
    ++x->n;
}

static int32_t run_error_value(const run_ctx* x, int32_t temp, int k) /* src/run_mode_context.hpp:86-99 */
{
    const int map = temp & 1;
    const int32_t ea = (temp + map) / 2;
    if ((k != 0 || (2 * x->nn >= x->n)) == map)
        return -ea;
    return ea;
}

/* ---------------------------------------------------------------- bit writer (src/scan_encoder.hpp:44-186) */

static void w_init(codec* c, uint8_t* dst, size_t size)
{
    c->wfree = 32;
    c->wbuf = 0;
    c->wpos = dst;
    c->wremaining = size;
    c->wwritten = 0;
    c->wff = 0;
}

static void w_flush(codec* c) /* src/scan_encoder.hpp:117-180 */
{
    if (c->wremaining < 4)
        fail(c, E_DESTINATION_TOO_SMALL);
    for (int i = 0; i < 4; ++i)
    {
        if (c->wfree >= 32)
        {
            c->wfree = 32;
            break;
        }
        uint8_t v;
        if (c->wff)
        {
            v = (uint8_t)(c->wbuf >> 25); /* 7 payload bits, stuffed 0 on top */
            c->wbuf <<= 7;
            c->wfree += 7;
        }
        else
        {
            v = (uint8_t)(c->wbuf >> 24);
            c->wbuf <<= 8;
            c->wfree += 8;
        }
        *c->wpos++ = v;
        c->wff = v == 0xFF;
        --c->wremaining;
        ++c->wwritten;
    }
}

static void w_append(codec* c, uint32_t bits, int count) /* src/scan_encoder.hpp:75-101 */
{
    c->wfree -= count;
    if (c->wfree >= 0)
    {
        c->wbuf |= count == 0 ? 0 : (bits << c->wfree);
        return;
    }
    c->wbuf |= bits >> -c->wfree;
    w_flush(c);
    if (c->wfree < 0)
    {
        c->wbuf |= bits >> -c->wfree;
        w_flush(c);
    }
    c->wbuf |= c->wfree >= 32 ? 0 : (bits << c->wfree);
}

static void w_end_scan(codec* c) /* src/scan_encoder.hpp:103-115 */
{
    w_flush(c);
    if (c->wff)
        w_append(c, 0, (c->wfree - 1) % 8);
    w_flush(c);
}

static void encode_mapped(codec* c, int k, int32_t m, int limit) /* src/scan_encoder_core.hpp:69-103 */
{
    int32_t hb = m >> k;
    if (hb < limit - c->qbpp - 1)
    {
        if (hb + 1 > 31)
        {
            w_append(c, 0, hb / 2);
            hb -= hb / 2;
        }
        const int total = hb + 1 + k;
        const uint32_t rem = (uint32_t)m & ((1u << k) - 1u);
        if (total < 32)
            w_append(c, (1u << k) | rem, total);
        else
        {
            w_append(c, 1, hb + 1);
            w_append(c, rem, k);
        }
        return;
    }
    if (limit - c->qbpp > 31)
    {
        w_append(c, 0, 31);
        w_append(c, 1, limit - c->qbpp - 31);
    }
    else
        w_append(c, 1, limit - c->qbpp);
    w_append(c, (uint32_t)(m - 1) & ((1u << c->qbpp) - 1u), c->qbpp);
}

/* ---------------------------------------------------------------- bit reader (src/scan_decoder.hpp:38-361) */

static void r_find_ff(codec* c) /* src/scan_decoder.hpp:324-333 */
{
    const uint8_t* p = c->rpos < c->rend ? memchr(c->rpos, 0xFF, (size_t)(c->rend - c->rpos)) : NULL;
    c->rff = p ? p : c->rend;
}

static void r_fill(codec* c) /* src/scan_decoder.hpp:252-322 */
{
    /* optimistic path: no 0xFF within the next 8 bytes */
    if (c->rff - c->rpos > 7 && c->rvalid >= 0)
    {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i)
            v = (v << 8) | c->rpos[i];
        c->rcache |= v >> c->rvalid;
        const int consumed = (64 - c->rvalid) / 8;
        c->rpos += consumed;
        c->rvalid += consumed * 8;
        return;
    }
    do
    {
        if (c->rpos >= c->rend)
        {
            if (c->rvalid <= 0) /* reference: == 0 (see header note on the documented deviation) */
                fail(c, E_INVALID_DATA);
            return;
        }
        const uint64_t b = *c->rpos;
        if (b == 0xFF && (c->rpos == c->rend - 1 || (c->rpos[1] & 0x80) != 0))
        {
            if (c->rvalid <= 0)
                fail(c, E_INVALID_DATA);
            return; /* marker (EOI, next SOS, RSTm) or end of buffer */
        }
        const int shift = 56 - c->rvalid;
        if (shift < 64)
            c->rcache |= b << shift;
        c->rvalid += 8;
        ++c->rpos;
        if (b == 0xFF)
            --c->rvalid; /* the stuffed bit after a data 0xFF is dropped */
    } while (c->rvalid < 56);
    r_find_ff(c);
}

static void r_init(codec* c, const uint8_t* src, size_t size)
{
    c->rpos = src;
    c->rend = src + size;
    c->rcache = 0;
    c->rvalid = 0;
    c->rst_counter = 0;
    r_find_ff(c);
    r_fill(c);
}

static void r_skip(codec* c, int n)
{
    c->rvalid -= n;
    c->rcache = n >= 64 ? 0 : (c->rcache << n);
}

static int32_t r_value(codec* c, int n) /* src/scan_decoder.hpp:127-142 */
{
    if (c->rvalid < n)
    {
        r_fill(c);
        if (c->rvalid < n)
            fail(c, E_INVALID_DATA);
    }
    const int32_t v = (int32_t)(c->rcache >> (64 - n));
    r_skip(c, n);
    return v;
}

static unsigned r_peek_byte(codec* c) /* src/scan_decoder.hpp:148-156 */
{
    if (c->rvalid < 8)
        r_fill(c);
    return (unsigned)(c->rcache >> 56);
}

static int r_bit(codec* c) /* src/scan_decoder.hpp:158-168 */
{
    if (c->rvalid <= 0)
        r_fill(c);
    const int bit = (int)(c->rcache >> 63);
    r_skip(c, 1);
    return bit;
}

static int32_t r_unary(codec* c) /* src/scan_decoder.hpp:176-217 */
{
    if (c->rvalid < 16)
        r_fill(c);
    int count = 0;
    while (count < 16 && !((c->rcache << count) >> 63))
        ++count;
    if (count < 16)
    {
        r_skip(c, count + 1);
        return count;
    }
    r_skip(c, 15);
    for (int32_t zeros = 15;; ++zeros)
        if (r_bit(c))
            return zeros;
}

static int32_t decode_mapped(codec* c, int k, int limit) /* src/scan_decoder.hpp:113-125 */
{
    const int32_t u = r_unary(c);
    if (u < limit - c->qbpp - 1)
        return k == 0 ? u : (u << k) + r_value(c, k);
    return r_value(c, c->qbpp) + 1;
}

static void r_end_scan(codec* c) /* src/scan_decoder.hpp:71-89 */
{
    if (c->rpos >= c->rend)
        fail(c, E_NEED_MORE_DATA);
    if (*c->rpos != 0xFF)
    {
        (void)r_bit(c);
        if (c->rpos >= c->rend)
            fail(c, E_NEED_MORE_DATA);
        if (*c->rpos != 0xFF)
            fail(c, E_INVALID_DATA);
    }
    if (c->rcache != 0)
        fail(c, E_INVALID_DATA);
}

static const uint8_t* r_actual_position(const codec* c) /* src/scan_decoder.hpp:92-107 */
{
    int valid = c->rvalid;
    const uint8_t* p = c->rpos;
    for (;;)
    {
        const int last = p[-1] == 0xFF ? 7 : 8;
        if (valid < last)
            return p;
        valid -= last;
        --p;
    }
}

static uint8_t r_byte(codec* c) /* src/scan_decoder.hpp:227-235 */
{
    if (c->rpos == c->rend)
        fail(c, E_NEED_MORE_DATA);
    return *c->rpos++;
}

static void r_restart_marker(codec* c) /* src/scan_decoder.hpp:237-243,335-349 */
{
    const uint32_t expected = 0xD0u + c->rst_counter;
    uint8_t v = r_byte(c);
    if (v != 0xFF)
        fail(c, E_RESTART_MARKER_NOT_FOUND);
    do
        v = r_byte(c);
    while (v == 0xFF);
    if (v != expected)
        fail(c, E_RESTART_MARKER_NOT_FOUND);
    c->rst_counter = (c->rst_counter + 1) % 8;
    c->rvalid = 0;
    c->rcache = 0;
    r_find_ff(c);
    r_fill(c);
}

/* ---------------------------------------------------------------- sample coding */

static int32_t qs_of(const codec* c, int32_t ra, int32_t rb, int32_t rc, int32_t rd) /* src/jpegls_algorithm.hpp:165-168 */
{
    return (c->q[rd - rb] * 9 + c->q[rb - rc]) * 9 + c->q[rc - ra];
}

static int32_t encode_regular(codec* c, int32_t qs, int32_t x, int32_t pred) /* src/scan_encoder_core.hpp:40-67 */
{
    const int32_t s = qs < 0 ? -1 : 0;
    reg_ctx* ctx = &c->reg[(qs ^ s) - s];
    const int k = reg_k(c, ctx);
    const int32_t px = correct_prediction(c, pred + ((ctx->c ^ s) - s));
    const int32_t e = compute_error_value(c, ((x - px) ^ s) - s);
    int32_t corr = 0;
    if ((k | c->near) == 0)
        corr = (2 * ctx->b + ctx->n - 1) < 0 ? -1 : 0; /* src/regular_mode_context.hpp:36-42 */
    encode_mapped(c, k, jls_oracle_map_error(corr ^ e), c->limit);
    reg_update(c, ctx, e);
    return reconstruct(c, px, (e ^ s) - s);
}

static int32_t decode_regular(codec* c, int32_t qs, int32_t pred) /* src/scan_decoder_core.hpp:38-69 */
{
    const int32_t s = qs < 0 ? -1 : 0;
    reg_ctx* ctx = &c->reg[(qs ^ s) - s];
    const int32_t px = correct_prediction(c, pred + ((ctx->c ^ s) - s));
    const int k = reg_k(c, ctx);
    int32_t e;
    const unsigned top = r_peek_byte(c);
    int u = 0;
    while (u < 8 && !((top << u) & 0x80))
        ++u;
    if (u + 1 + k <= 8)
    { /* golomb_lut hit: the whole code lies in the first 8 bits (src/golomb_lut.cpp:24-63) */
        const int32_t m = (u << k) | (int32_t)((top >> (8 - u - 1 - k)) & ((1u << k) - 1u));
        r_skip(c, u + 1 + k);
        e = jls_oracle_unmap_error(m);
    }
    else
    {
        e = jls_oracle_unmap_error(decode_mapped(c, k, c->limit));
        if (e > 65535 || e < -65535)
            fail(c, E_INVALID_DATA);
    }
    if (k == 0 && c->near == 0)
        e ^= (2 * ctx->b + ctx->n - 1) < 0 ? -1 : 0;
    reg_update(c, ctx, e);
    return reconstruct(c, px, (e ^ s) - s);
}

static void encode_ri_error(codec* c, run_ctx* ctx, int32_t e) /* src/scan_encoder_core.hpp:105-116 */
{
    const int k = run_k(c, ctx, 0);
    const int map = run_map(ctx, e, k);
    const int32_t em = 2 * (e < 0 ? -e : e) - ctx->ritype - map;
    encode_mapped(c, k, em, c->limit - J[c->run_index] - 1);
    run_update(c, ctx, e, em);
}

static int32_t decode_ri_error(codec* c, run_ctx* ctx) /* src/scan_decoder_core.hpp:72-81 */
{
    const int k = run_k(c, ctx, 1);
    const int32_t em = decode_mapped(c, k, c->limit - J[c->run_index] - 1);
    const int32_t e = run_error_value(ctx, em + ctx->ritype, k);
    run_update(c, ctx, e, em);
    return e;
}

static void encode_run_pixels(codec* c, size_t run_length, int eol) /* src/scan_encoder.hpp:53-73 */
{
    while (run_length >= ((size_t)1 << J[c->run_index]))
    {
        w_append(c, 1, 1);
        run_length -= (size_t)1 << J[c->run_index];
        if (c->run_index < 31)
            ++c->run_index;
    }
    if (eol)
    {
        if (run_length != 0)
            w_append(c, 1, 1);
    }
    else
        w_append(c, (uint32_t)run_length, J[c->run_index] + 1);
}

static size_t decode_run_pixels(codec* c, size_t remaining) /* src/scan_decoder_impl.hpp:301-337 (fill is done by caller) */
{
    size_t index = 0;
    while (r_bit(c))
    {
        const size_t block = (size_t)1 << J[c->run_index];
        const size_t count = block < remaining - index ? block : remaining - index;
        index += count;
        if (count == block && c->run_index < 31)
            ++c->run_index;
        if (index == remaining)
            break;
    }
    if (index != remaining)
        index += J[c->run_index] > 0 ? (size_t)r_value(c, J[c->run_index]) : 0;
    if (index > remaining)
        fail(c, E_INVALID_DATA);
    return index;
}

/* One line of one scan. prev/cur point at element 0 of component 0's (w+2)-sample sub-line; component j lives at
 * offset j*stride. `nc` > 1 only for ILV_SAMPLE (src/scan_encoder_impl.hpp:109-302, src/scan_decoder_impl.hpp:132-337). */
static void code_line(codec* c, uint16_t* prev, uint16_t* cur, int nc, size_t stride, int decode)
{
    const uint32_t w = c->w;
    uint32_t i = 1;
    while (i <= w)
    {
        int32_t qs[4];
        int all_zero = 1;
        for (int j = 0; j < nc; ++j)
        {
            const uint16_t* p = prev + j * stride;
            const uint16_t* q = cur + j * stride;
            qs[j] = qs_of(c, q[i - 1], p[i], p[i - 1], p[i + 1]);
            all_zero &= qs[j] == 0;
        }
        if (!all_zero)
        {
            for (int j = 0; j < nc; ++j)
            {
                const uint16_t* p = prev + j * stride;
                uint16_t* q = cur + j * stride;
                const int32_t pred = jls_oracle_med(q[i - 1], p[i], p[i - 1]);
                q[i] = (uint16_t)(decode ? decode_regular(c, qs[j], pred) : encode_regular(c, qs[j], q[i], pred));
            }
            ++i;
            continue;
        }
        /* run mode */
        const uint32_t remaining = w - (i - 1);
        uint32_t run = 0;
        if (decode)
        {
            run = (uint32_t)decode_run_pixels(c, remaining);
            for (int j = 0; j < nc; ++j)
            {
                uint16_t* q = cur + j * stride;
                for (uint32_t t = 0; t < run; ++t)
                    q[i + t] = q[i - 1];
            }
        }
        else
        {
            for (;;)
            {
                int near_all = 1;
                for (int j = 0; j < nc; ++j)
                {
                    const uint16_t* q = cur + j * stride;
                    near_all &= is_near(c, q[i + run], q[i - 1]);
                }
                if (!near_all)
                    break;
                for (int j = 0; j < nc; ++j)
                {
                    uint16_t* q = cur + j * stride;
                    q[i + run] = q[i - 1];
                }
                if (++run == remaining)
                    break;
            }
            encode_run_pixels(c, run, run == remaining);
        }
        if (run == remaining)
            break;
        /* run interruption sample at i + run */
        const uint32_t e_idx = i + run;
        for (int j = 0; j < nc; ++j)
        {
            const uint16_t* p = prev + j * stride;
            uint16_t* q = cur + j * stride;
            const int32_t ra = q[i - 1];
            const int32_t rb = p[e_idx];
            int32_t rx;
            if (nc == 1 && is_near(c, ra, rb))
            { /* src/scan_encoder_core.hpp:118-125, src/scan_decoder_core.hpp:84-90 */
                int32_t e;
                if (decode)
                    e = decode_ri_error(c, &c->rc[1]);
                else
                {
                    e = compute_error_value(c, q[e_idx] - ra);
                    encode_ri_error(c, &c->rc[1], e);
                }
                rx = reconstruct(c, ra, e);
            }
            else
            { /* src/scan_encoder_core.hpp:127-138, src/scan_decoder_core.hpp:92-100 */
                const int32_t sg = sgn(rb - ra);
                int32_t e;
                if (decode)
                    e = decode_ri_error(c, &c->rc[0]);
                else
                {
                    e = compute_error_value(c, (q[e_idx] - rb) * sg);
                    encode_ri_error(c, &c->rc[0], e);
                }
                rx = reconstruct(c, rb, e * sg);
            }
            q[e_idx] = (uint16_t)rx;
        }
        if (c->run_index > 0)
            --c->run_index;
        i = e_idx + 1;
    }
}

/* ---------------------------------------------------------------- pixel conversion */

static uint32_t load_sample(const uint8_t* p, int wide)
{
    return wide ? (uint32_t)(p[0] | (p[1] << 8)) : p[0];
}

static void store_sample(uint8_t* p, int wide, uint32_t v)
{
    p[0] = (uint8_t)v;
    if (wide)
        p[1] = (uint8_t)(v >> 8);
}

/* src/color_transform.hpp:26-117; all arithmetic modulo 2^(8*sizeof(sample)) */
static void hp_forward(int xform, int wide, int32_t r, int32_t g, int32_t b, uint32_t out[3])
{
    const int32_t range = wide ? 65536 : 256;
    const int32_t bias = range / 2;
    const uint32_t m = (uint32_t)range - 1u;
    if (xform == 1)
    {
        out[0] = (uint32_t)(r - g + bias) & m;
        out[1] = (uint32_t)g & m;
        out[2] = (uint32_t)(b - g + bias) & m;
    }
    else if (xform == 2)
    {
        out[0] = (uint32_t)(r - g + bias) & m;
        out[1] = (uint32_t)g & m;
        out[2] = (uint32_t)(b - ((r + g) / 2) + bias) & m;
    }
    else
    {
        const int32_t v2 = (int32_t)((uint32_t)(b - g + bias) & m);
        const int32_t v3 = (int32_t)((uint32_t)(r - g + bias) & m);
        out[0] = (uint32_t)(g + ((v2 + v3) >> 2) - range / 4) & m;
        out[1] = (uint32_t)v2;
        out[2] = (uint32_t)v3;
    }
}

static void hp_inverse(int xform, int wide, int32_t v1, int32_t v2, int32_t v3, uint32_t out[3])
{
    const int32_t range = wide ? 65536 : 256;
    const int32_t bias = range / 2;
    const uint32_t m = (uint32_t)range - 1u;
    if (xform == 1)
    {
        out[0] = (uint32_t)(v1 + v2 - bias) & m;
        out[1] = (uint32_t)v2 & m;
        out[2] = (uint32_t)(v3 + v2 - bias) & m;
    }
    else if (xform == 2)
    {
        const int32_t r = (int32_t)((uint32_t)(v1 + v2 - bias) & m);
        out[0] = (uint32_t)r;
        out[1] = (uint32_t)v2 & m;
        out[2] = (uint32_t)(v3 + ((r + (int32_t)((uint32_t)v2 & m)) >> 1) - bias) & m;
    }
    else
    {
        const int32_t g = v1 - ((v3 + v2) >> 2) + range / 4;
        out[0] = (uint32_t)(v3 + g - bias) & m;
        out[1] = (uint32_t)g & m;
        out[2] = (uint32_t)(v2 + g - bias) & m;
    }
}

/* src/copy_to_line_buffer.hpp:21-262: user row -> planar (w+2)-strided sub-lines, element 1.. */
static void row_to_lines(const codec* c, const uint8_t* src, uint16_t* cur, size_t stride)
{
    const int wide = c->bpp > 8;
    const int bytes = wide ? 2 : 1;
    const uint32_t mask = (1u << c->bpp) - 1u;
    if (c->ilv == 0)
    {
        const int need_mask = c->bpp != bytes * 8;
        for (uint32_t i = 0; i < c->w; ++i)
        {
            const uint32_t v = load_sample(src + (size_t)i * bytes, wide);
            cur[1 + i] = (uint16_t)(need_mask ? v & mask : v);
        }
        return;
    }
    for (uint32_t i = 0; i < c->w; ++i)
    {
        uint32_t v[4];
        for (int j = 0; j < c->nc; ++j)
            v[j] = load_sample(src + ((size_t)i * c->nc + j) * bytes, wide);
        if (c->xform != 0 && c->nc == 3)
            hp_forward(c->xform, wide, (int32_t)v[0], (int32_t)v[1], (int32_t)v[2], v);
        else
            for (int j = 0; j < c->nc; ++j)
                v[j] &= mask;
        for (int j = 0; j < c->nc; ++j)
            cur[j * stride + 1 + i] = (uint16_t)v[j];
    }
}

/* src/copy_from_line_buffer.hpp:19-191 */
static void lines_to_row(const codec* c, const uint16_t* cur, size_t stride, uint8_t* dst)
{
    const int wide = c->bpp > 8;
    const int bytes = wide ? 2 : 1;
    if (c->ilv == 0)
    {
        for (uint32_t i = 0; i < c->w; ++i)
            store_sample(dst + (size_t)i * bytes, wide, cur[1 + i]);
        return;
    }
    for (uint32_t i = 0; i < c->w; ++i)
    {
        uint32_t v[4];
        for (int j = 0; j < c->nc; ++j)
            v[j] = cur[j * stride + 1 + i];
        if (c->xform != 0 && c->nc == 3)
            hp_inverse(c->xform, wide, (int32_t)v[0], (int32_t)v[1], (int32_t)v[2], v);
        for (int j = 0; j < c->nc; ++j)
            store_sample(dst + ((size_t)i * c->nc + j) * bytes, wide, v[j]);
    }
}

/* ---------------------------------------------------------------- scan set-up */

static int scan_setup(codec* c, const jls_oracle_params* p)
{
    memset(c, 0, sizeof *c);
    c->w = p->width;
    c->h = p->height;
    c->nc = p->component_count;
    c->ilv = p->interleave_mode;
    c->near = p->near_lossless;
    c->bpp = p->bits_per_sample;
    c->xform = p->color_transformation;
    /* src/make_scan_codec.cpp:40-156: every path uses MAXVAL = 2^bpp - 1 for RANGE/qbpp/LIMIT */
    c->maxval = (1 << c->bpp) - 1;
    c->range = (c->maxval + 2 * c->near) / (2 * c->near + 1) + 1; /* src/jpegls_algorithm.hpp:124-130 */
    c->qbpp = log2_ceiling(c->range);
    c->limit = 2 * (c->bpp + imax(8, c->bpp)); /* src/jpegls_algorithm.hpp:137-140 */
    c->t1 = p->threshold1;
    c->t2 = p->threshold2;
    c->t3 = p->threshold3;
    c->reset = (uint8_t)p->reset_value; /* src/scan_codec.hpp:142: stored through a uint8_t cast */
    c->restart_interval = p->restart_interval;

    const size_t qn = (size_t)1 << c->bpp; /* src/scan_codec.hpp:89-98 */
    c->qstore = malloc(2 * qn);
    const size_t planes = c->ilv == 0 ? 1 : (size_t)c->nc;
    c->lines = calloc(2 * planes * ((size_t)c->w + 2), sizeof(uint16_t));
    if (!c->qstore || !c->lines)
        return E_NOT_ENOUGH_MEMORY;
    for (size_t i = 0; i < 2 * qn; ++i)
        c->qstore[i] = (int8_t)jls_oracle_quantize_gradient((int32_t)i - (int32_t)qn, c->t1, c->t2, c->t3, c->near);
    c->q = c->qstore + qn;
    init_model(c);
    return E_OK;
}

static void scan_free(codec* c)
{
    free(c->qstore);
    free(c->lines);
}

/* src/scan_encoder_impl.hpp:55-106 */
static void encode_lines(codec* c, const uint8_t* src, size_t stride_bytes)
{
    const size_t ps = (size_t)c->w + 2;
    const size_t planes = c->ilv == 0 ? 1 : (size_t)c->nc;
    int run_index[4] = {0, 0, 0, 0};
    for (uint32_t line = 0; line < c->h; ++line)
    {
        uint16_t* prev = c->lines;
        uint16_t* cur = c->lines + planes * ps;
        if (line & 1)
        {
            uint16_t* t = prev;
            prev = cur;
            cur = t;
        }
        row_to_lines(c, src, cur, ps);
        src += stride_bytes;
        if (c->ilv == 2)
        {
            for (int j = 0; j < c->nc; ++j)
            { /* src/scan_codec.hpp:189-195 on whole pixels */
                prev[j * ps + c->w + 1] = prev[j * ps + c->w];
                cur[j * ps] = prev[j * ps + 1];
            }
            c->run_index = run_index[0];
            code_line(c, prev, cur, c->nc, ps, 0);
            run_index[0] = c->run_index;
        }
        else
        {
            for (size_t j = 0; j < planes; ++j)
            {
                uint16_t* p = prev + j * ps;
                uint16_t* q = cur + j * ps;
                c->run_index = run_index[j];
                p[c->w + 1] = p[c->w];
                q[0] = p[1];
                code_line(c, p, q, 1, ps, 0);
                run_index[j] = c->run_index;
            }
        }
    }
}

/* src/scan_decoder_impl.hpp:62-129 */
static void decode_lines(codec* c, uint8_t* dst, size_t stride_bytes)
{
    const size_t ps = (size_t)c->w + 2;
    const size_t planes = c->ilv == 0 ? 1 : (size_t)c->nc;
    int run_index[4] = {0, 0, 0, 0};
    uint32_t interval = c->restart_interval == 0 ? c->h : c->restart_interval;
    for (uint32_t line = 0;;)
    {
        const uint32_t lines_in_interval = (c->h - line) < interval ? (c->h - line) : interval;
        for (uint32_t mcu = 0; mcu < lines_in_interval; ++mcu, ++line)
        {
            uint16_t* prev = c->lines;
            uint16_t* cur = c->lines + planes * ps;
            if (line & 1)
            {
                uint16_t* t = prev;
                prev = cur;
                cur = t;
            }
            if (c->ilv == 2)
            {
                for (int j = 0; j < c->nc; ++j)
                {
                    prev[j * ps + c->w + 1] = prev[j * ps + c->w];
                    cur[j * ps] = prev[j * ps + 1];
                }
                c->run_index = run_index[0];
                code_line(c, prev, cur, c->nc, ps, 1);
                run_index[0] = c->run_index;
            }
            else
            {
                for (size_t j = 0; j < planes; ++j)
                {
                    uint16_t* p = prev + j * ps;
                    uint16_t* q = cur + j * ps;
                    c->run_index = run_index[j];
                    p[c->w + 1] = p[c->w];
                    q[0] = p[1];
                    code_line(c, p, q, 1, ps, 1);
                    run_index[j] = c->run_index;
                }
            }
            lines_to_row(c, cur, ps, dst);
            dst += stride_bytes;
        }
        if (line == c->h)
            break;
        r_restart_marker(c);
        memset(run_index, 0, sizeof run_index);
        memset(c->lines, 0, 2 * planes * ps * sizeof(uint16_t));
        init_model(c);
    }
}

int jls_oracle_encode_scan(const jls_oracle_params* params, const void* source, size_t stride, void* destination,
                           size_t destination_size, size_t* bytes_written)
{
    codec* c = malloc(sizeof *c);
    if (!c)
        return E_NOT_ENOUGH_MEMORY;
    int rc = scan_setup(c, params);
    if (rc == E_OK)
    {
        rc = setjmp(c->fail);
        if (rc == 0)
        {
            w_init(c, destination, destination_size);
            encode_lines(c, source, stride);
            w_end_scan(c);
            *bytes_written = c->wwritten;
        }
    }
    scan_free(c);
    free(c);
    return rc;
}

int jls_oracle_decode_scan(const jls_oracle_params* params, const void* source, size_t source_size, void* destination,
                           size_t stride, size_t* bytes_read)
{
    codec* c = malloc(sizeof *c);
    if (!c)
        return E_NOT_ENOUGH_MEMORY;
    int rc = scan_setup(c, params);
    if (rc == E_OK)
    {
        rc = setjmp(c->fail);
        if (rc == 0)
        {
            r_init(c, source, source_size);
            decode_lines(c, destination, stride);
            r_end_scan(c);
            *bytes_read = (size_t)(r_actual_position(c) - (const uint8_t*)source);
        }
    }
    scan_free(c);
    free(c);
    return rc;
}

int jls_oracle_bitwriter_kat(const uint32_t* values, const int32_t* bit_counts, int count, uint8_t* destination,
                             size_t destination_size, size_t* bytes_written)
{
    codec* c = calloc(1, sizeof *c);
    if (!c)
        return E_NOT_ENOUGH_MEMORY;
    int rc = setjmp(c->fail);
    if (rc == 0)
    {
        w_init(c, destination, destination_size);
        for (int i = 0; i < count; ++i)
            w_append(c, values[i], bit_counts[i]);
        w_end_scan(c);
        *bytes_written = c->wwritten;
    }
    free(c);
    return rc;
}

/* ---------------------------------------------------------------- container: writer (src/jpeg_stream_writer.cpp) */

typedef struct
{
    uint8_t* p;
    size_t cap, off;
    int err;
} wr;

static void wr_u8(wr* w, unsigned v)
{
    w->p[w->off++] = (uint8_t)v;
}
static void wr_u16(wr* w, unsigned v)
{
    wr_u8(w, v >> 8);
    wr_u8(w, v);
}
static void wr_u32(wr* w, uint32_t v)
{
    wr_u16(w, v >> 16);
    wr_u16(w, v & 0xffff);
}
static int wr_segment(wr* w, unsigned marker, size_t data_size) /* src/jpeg_stream_writer.cpp:229-243 */
{
    if (w->off + 4 + data_size > w->cap)
    {
        w->err = E_DESTINATION_TOO_SMALL;
        return 0;
    }
    wr_u8(w, 0xFF);
    wr_u8(w, marker);
    wr_u16(w, (unsigned)(2 + data_size));
    return 1;
}
static int wr_marker(wr* w, unsigned marker) /* src/jpeg_stream_writer.hpp write_segment_without_data */
{
    if (w->off + 2 > w->cap)
    {
        w->err = E_DESTINATION_TOO_SMALL;
        return 0;
    }
    wr_u8(w, 0xFF);
    wr_u8(w, marker);
    return 1;
}

static int color_transform_possible(const jls_oracle_params* p) /* src/color_transform.hpp:11-16 */
{
    return p->component_count == 3 && (p->bits_per_sample == 8 || p->bits_per_sample == 16) && p->near_lossless == 0 &&
           p->interleave_mode != 0;
}

int jls_oracle_encode(const jls_oracle_params* p, const void* source, size_t source_size, uint32_t stride_arg,
                      void* destination, size_t destination_size, size_t* bytes_written)
{ /* src/charls_jpegls_encoder.cpp:44-53,182-236,285-424 */
    if (!source || !destination || !bytes_written)
        return E_INVALID_ARGUMENT;
    if (p->width < 1 || p->width > 100000)
        return E_INVALID_ARGUMENT_WIDTH;
    if (p->height < 1 || p->height > 100000)
        return E_INVALID_ARGUMENT_HEIGHT;
    if (p->bits_per_sample < 2 || p->bits_per_sample > 16)
        return E_INVALID_ARGUMENT_BPS;
    if (p->component_count < 1 || p->component_count > 255)
        return E_INVALID_ARGUMENT_COMPONENT_COUNT;
    if (p->interleave_mode < 0 || p->interleave_mode > 2)
        return E_INVALID_ARGUMENT_ILV;
    if (p->near_lossless < 0 || p->near_lossless > 255)
        return E_INVALID_ARGUMENT_NEAR;
    if (p->color_transformation < 0 || p->color_transformation > 3)
        return E_INVALID_ARGUMENT_COLOR_TRANSFORMATION;
    if (p->component_count == 1 && p->interleave_mode != 0)
        return E_INVALID_ARGUMENT_ILV;
    if (p->interleave_mode != 0 && p->component_count > 4)
        return E_INVALID_ARGUMENT_ILV; /* src/util.hpp check_interleave_mode is value-only; >4 asserts in the reference */

    const int32_t bit_maxval = (1 << p->bits_per_sample) - 1;
    const int32_t user_pc[5] = {p->maximum_sample_value, p->threshold1, p->threshold2, p->threshold3, p->reset_value};
    int32_t maxval_for_near = bit_maxval;
    if (user_pc[0] != 0)
    {
        if (user_pc[0] < 1 || user_pc[0] > bit_maxval)
            return E_INVALID_ARGUMENT_PC;
        maxval_for_near = user_pc[0];
    }
    if (p->near_lossless > imin(255, maxval_for_near / 2))
        return E_INVALID_ARGUMENT_NEAR;

    const size_t bytes = (size_t)((p->bits_per_sample + 7) / 8);
    const size_t min_stride = (size_t)p->width * bytes * (p->interleave_mode == 0 ? 1 : (size_t)p->component_count);
    size_t stride = stride_arg;
    if (stride == 0)
        stride = min_stride;
    else if (stride < min_stride)
        return E_INVALID_ARGUMENT_STRIDE;
    const size_t unused = stride - min_stride;
    const size_t min_size = (p->interleave_mode == 0 ? stride * (size_t)p->component_count * p->height : stride * p->height) - unused;
    if (source_size < min_size)
        return E_INVALID_ARGUMENT_SIZE;

    int32_t pc[5];
    if (!pc_validate(user_pc, bit_maxval, p->near_lossless, pc))
        return E_INVALID_ARGUMENT_PC;

    wr w = {destination, destination_size, 0, 0};
    if (!wr_marker(&w, 0xD8))
        return w.err;
    if (p->encoding_options & 2u)
    { /* src/charls_jpegls_encoder.cpp:378-383: "charls 3.0.0" + NUL */
        static const char version[] = "charls 3.0.0";
        if (!wr_segment(&w, 0xFE, sizeof version))
            return w.err;
        memcpy(w.p + w.off, version, sizeof version);
        w.off += sizeof version;
    }
    if (p->color_transformation != 0)
    {
        if (!color_transform_possible(p))
            return E_INVALID_ARGUMENT_COLOR_TRANSFORMATION;
        if (!wr_segment(&w, 0xE8, 5))
            return w.err;
        wr_u8(&w, 'm');
        wr_u8(&w, 'r');
        wr_u8(&w, 'f');
        wr_u8(&w, 'x');
        wr_u8(&w, (unsigned)p->color_transformation);
    }
    { /* SOF55, src/jpeg_stream_writer.cpp:83-114 */
        if (!wr_segment(&w, 0xF7, 6 + (size_t)p->component_count * 3))
            return w.err;
        const int oversize = p->width > 65535 || p->height > 65535;
        wr_u8(&w, (unsigned)p->bits_per_sample);
        wr_u16(&w, oversize ? 0 : p->height);
        wr_u16(&w, oversize ? 0 : p->width);
        wr_u8(&w, (unsigned)p->component_count);
        for (int i = 1; i <= p->component_count; ++i)
        {
            wr_u8(&w, (unsigned)i);
            wr_u8(&w, 0x11);
            wr_u8(&w, 0);
        }
        if (oversize)
        { /* src/jpeg_stream_writer.cpp:153-161 */
            if (!wr_segment(&w, 0xF8, 10))
                return w.err;
            wr_u8(&w, 4);
            wr_u8(&w, 4);
            wr_u32(&w, p->height);
            wr_u32(&w, p->width);
        }
    }
    { /* LSE type 1, src/charls_jpegls_encoder.cpp:409-418 */
        int32_t d[5];
        jls_oracle_default_pc(bit_maxval, p->near_lossless, d);
        if (!pc_is_default(user_pc, d) || ((p->encoding_options & 4u) && p->bits_per_sample > 12))
        {
            if (!wr_segment(&w, 0xF8, 11))
                return w.err;
            wr_u8(&w, 1);
            for (int i = 0; i < 5; ++i)
                wr_u16(&w, (unsigned)pc[i]);
        }
    }

    jls_oracle_params sp = *p;
    sp.threshold1 = pc[1];
    sp.threshold2 = pc[2];
    sp.threshold3 = pc[3];
    sp.reset_value = pc[4];
    sp.restart_interval = 0;
    const uint8_t* src = source;
    const int scans = p->interleave_mode == 0 ? p->component_count : 1;
    const int comps_per_scan = p->interleave_mode == 0 ? 1 : p->component_count;
    int component_id = 1;
    for (int s = 0; s < scans; ++s)
    { /* SOS, src/jpeg_stream_writer.cpp:185-207 */
        if (!wr_segment(&w, 0xDA, 1 + (size_t)comps_per_scan * 2 + 3))
            return w.err;
        wr_u8(&w, (unsigned)comps_per_scan);
        for (int i = 0; i < comps_per_scan; ++i)
        {
            wr_u8(&w, (unsigned)component_id++);
            wr_u8(&w, 0);
        }
        wr_u8(&w, (unsigned)p->near_lossless);
        wr_u8(&w, (unsigned)p->interleave_mode);
        wr_u8(&w, 0);
        sp.component_count = comps_per_scan;
        size_t n = 0;
        const int rc = jls_oracle_encode_scan(&sp, src, stride, w.p + w.off, w.cap - w.off, &n);
        if (rc != E_OK)
            return rc;
        w.off += n;
        src += stride * p->height;
    }
    if ((p->encoding_options & 1u) && (w.off % 2) != 0)
    { /* src/jpeg_stream_writer.cpp:26-35 */
        if (w.off + 1 > w.cap)
            return E_DESTINATION_TOO_SMALL;
        wr_u8(&w, 0xFF);
    }
    if (!wr_marker(&w, 0xD9))
        return w.err;
    *bytes_written = w.off;
    return E_OK;
}

/* ---------------------------------------------------------------- container: reader (subset of src/jpeg_stream_reader.cpp) */

typedef struct
{
    const uint8_t* p;
    const uint8_t* end;
    jls_oracle_params prm;
    int32_t pc[5];
    uint8_t comp_ids[255];
    int read_components;
    int scan_components;
    int state; /* 0 header, 1 after SOF (scan section), 2 bit stream */
} rd;

static int rd_marker(rd* r, unsigned* marker) /* src/jpeg_stream_reader.cpp:192-213 */
{
    if (r->p == r->end)
        return E_NEED_MORE_DATA;
    if (*r->p++ != 0xFF)
        return E_MARKER_START_BYTE_NOT_FOUND;
    unsigned m;
    do
    {
        if (r->p == r->end)
            return E_NEED_MORE_DATA;
        m = *r->p++;
    } while (m == 0xFF);
    *marker = m;
    return E_OK;
}

static unsigned be16(const uint8_t* p)
{
    return (unsigned)(p[0] << 8 | p[1]);
}
static uint32_t be_n(const uint8_t* p, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; ++i)
        v = (v << 8) | p[i];
    return v;
}

static int rd_set_height(rd* r, uint32_t h) /* src/jpeg_stream_reader.cpp:898-907 */
{
    if (h == 0)
        return E_OK;
    if (r->prm.height != 0 || h > 100000)
        return E_INVALID_PARAMETER_HEIGHT;
    r->prm.height = h;
    return E_OK;
}
static int rd_set_width(rd* r, uint32_t w) /* src/jpeg_stream_reader.cpp:910-919 */
{
    if (w == 0)
        return E_OK;
    if (r->prm.width != 0 || w > 100000)
        return E_INVALID_PARAMETER_WIDTH;
    r->prm.width = w;
    return E_OK;
}

/* Reads marker segments until the next SOS has been consumed. */
static int rd_until_scan(rd* r)
{
    for (;;)
    {
        unsigned m;
        int rc = rd_marker(r, &m);
        if (rc)
            return rc;
        if (m == 0xD9)
            return E_UNEXPECTED_EOI;
        if (m == 0xD8)
            return E_DUPLICATE_SOI;
        if (m == 0xDA && r->state != 1)
            return E_UNEXPECTED_SOS;
        if (m == 0xF7 && r->state == 1)
            return E_DUPLICATE_SOF;
        const int known = m == 0xDA || m == 0xF7 || m == 0xF8 || m == 0xDD || m == 0xFE || (m >= 0xE0 && m <= 0xEF);
        if (!known)
        {
            if (m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC3 || m == 0xC5 || m == 0xC6 || m == 0xC7 || m == 0xC9 || m == 0xCA ||
                m == 0xCB || m == 0xF9)
                return E_ENCODING_NOT_SUPPORTED;
            if (m >= 0xD0 && m <= 0xD7)
                return E_UNEXPECTED_RESTART_MARKER;
            return E_UNKNOWN_MARKER;
        }
        if (r->p + 2 > r->end)
            return E_NEED_MORE_DATA;
        const size_t seg = be16(r->p);
        r->p += 2;
        if (seg < 2 || r->p + (seg - 2) > r->end)
            return E_INVALID_SEGMENT_SIZE;
        const uint8_t* d = r->p;
        const size_t n = seg - 2;
        r->p += n;
        switch (m)
        {
        case 0xF7: /* src/jpeg_stream_reader.cpp:375-409 */
            if (n < 6)
                return E_INVALID_SEGMENT_SIZE;
            r->prm.bits_per_sample = d[0];
            if (d[0] < 2 || d[0] > 16)
                return E_INVALID_PARAMETER_BPS;
            if ((rc = rd_set_height(r, be16(d + 1))) != 0 || (rc = rd_set_width(r, be16(d + 3))) != 0)
                return rc;
            r->prm.component_count = d[5];
            if (d[5] == 0)
                return E_INVALID_PARAMETER_COMPONENT_COUNT;
            if (n != (size_t)d[5] * 3 + 6)
                return E_INVALID_SEGMENT_SIZE;
            for (int i = 0; i < d[5]; ++i)
            {
                for (int j = 0; j < i; ++j)
                    if (r->comp_ids[j] == d[6 + i * 3])
                        return E_DUPLICATE_COMPONENT_ID;
                r->comp_ids[i] = d[6 + i * 3];
                if (d[7 + i * 3] != 0x11)
                    return E_PARAMETER_VALUE_NOT_SUPPORTED;
            }
            r->state = 1;
            break;
        case 0xF8: /* src/jpeg_stream_reader.cpp:475-558 */
            if (n < 1)
                return E_INVALID_SEGMENT_SIZE;
            if (d[0] == 1)
            {
                if (n != 11)
                    return E_INVALID_SEGMENT_SIZE;
                for (int i = 0; i < 5; ++i)
                    r->pc[i] = (int32_t)be16(d + 1 + 2 * i);
            }
            else if (d[0] == 4)
            {
                if (n < 2)
                    return E_INVALID_SEGMENT_SIZE;
                const int sz = d[1];
                if (sz < 2 || sz > 4)
                    return E_INVALID_PARAMETER_PC;
                if (n != 2 + 2 * (size_t)sz)
                    return E_INVALID_SEGMENT_SIZE;
                if ((rc = rd_set_height(r, be_n(d + 2, sz))) != 0 || (rc = rd_set_width(r, be_n(d + 2 + sz, sz))) != 0)
                    return rc;
            }
            else if (d[0] == 2 || d[0] == 3)
            {
                /* mapping tables: not interpreted by the oracle */
            }
            else
                return d[0] <= 0xD ? E_PRESET_EXTENDED_NOT_SUPPORTED : E_INVALID_PRESET_TYPE;
            break;
        case 0xDD: /* src/jpeg_stream_reader.cpp:586-607 */
            if (n < 2 || n > 4)
                return E_INVALID_SEGMENT_SIZE;
            r->prm.restart_interval = be_n(d, (int)n);
            break;
        case 0xE8: /* src/jpeg_stream_reader.cpp:764-809 */
            if (n == 5 && memcmp(d, "mrfx", 4) == 0)
            {
                if (d[4] <= 3)
                    r->prm.color_transformation = d[4];
                else if (d[4] == 4 || d[4] == 5)
                    return E_COLOR_TRANSFORM_NOT_SUPPORTED;
                else
                    return E_INVALID_PARAMETER_COLOR_TRANSFORMATION;
            }
            break;
        case 0xDA: /* src/jpeg_stream_reader.cpp:610-654 */
        {
            if (n < 1)
                return E_INVALID_SEGMENT_SIZE;
            const int sc = d[0];
            if (sc < 1 || sc > 4 || sc > r->prm.component_count - r->read_components)
                return E_INVALID_PARAMETER_COMPONENT_COUNT;
            r->scan_components = sc;
            r->read_components += sc;
            if (n != (size_t)sc * 2 + 4)
                return E_INVALID_SEGMENT_SIZE;
            const int near = d[1 + sc * 2];
            const int32_t mv = r->pc[0] != 0 ? r->pc[0] : (1 << r->prm.bits_per_sample) - 1;
            if (near > imin(255, mv / 2))
                return E_INVALID_PARAMETER_NEAR;
            r->prm.near_lossless = near;
            const int ilv = d[2 + sc * 2];
            if (ilv > 2 || (sc == 1 && ilv != 0))
                return E_INVALID_PARAMETER_ILV;
            r->prm.interleave_mode = ilv;
            if ((d[3 + sc * 2] & 0xF) != 0)
                return E_PARAMETER_VALUE_NOT_SUPPORTED;
            r->state = 2;
            return E_OK;
        }
        default:
            break; /* COM / APPn: skipped */
        }
    }
}

static int rd_header(rd* r, const void* source, size_t size)
{
    memset(r, 0, sizeof *r);
    r->p = source;
    r->end = r->p + size;
    unsigned m;
    int rc = rd_marker(r, &m);
    if (rc)
        return rc;
    if (m != 0xD8)
        return E_SOI_NOT_FOUND;
    rc = rd_until_scan(r);
    if (rc)
        return rc;
    if (r->prm.height == 0)
        return 26; /* define_number_of_lines_marker_not_found: DNL is not supported by the oracle */
    if (r->prm.width < 1)
        return E_INVALID_PARAMETER_WIDTH;
    if (r->prm.color_transformation != 0 && !color_transform_possible(&r->prm))
        return E_INVALID_PARAMETER_COLOR_TRANSFORMATION;
    r->prm.maximum_sample_value = r->pc[0];
    r->prm.threshold1 = r->pc[1];
    r->prm.threshold2 = r->pc[2];
    r->prm.threshold3 = r->pc[3];
    r->prm.reset_value = r->pc[4];
    return E_OK;
}

int jls_oracle_read_header(const void* source, size_t source_size, jls_oracle_params* params)
{
    rd r;
    const int rc = rd_header(&r, source, source_size);
    if (rc == E_OK)
        *params = r.prm;
    return rc;
}

int jls_oracle_decode(const void* source, size_t source_size, void* destination, size_t destination_size,
                      uint32_t stride_arg, jls_oracle_params* params_out)
{ /* src/charls_jpegls_decoder.cpp:177-247 */
    rd r;
    int rc = rd_header(&r, source, source_size);
    if (rc)
        return rc;
    const size_t bytes = (size_t)((r.prm.bits_per_sample + 7) / 8);
    uint8_t* dst = destination;
    size_t remaining = destination_size;
    for (int component = 0;;)
    {
        const size_t planes = r.prm.interleave_mode == 0 ? 1 : (size_t)r.scan_components;
        const size_t min_stride = planes * r.prm.width * bytes;
        size_t stride = stride_arg;
        if (stride == 0)
            stride = min_stride;
        else if (stride < min_stride)
            return E_INVALID_ARGUMENT_STRIDE;
        const size_t unused = stride - min_stride;
        const size_t need = (r.prm.interleave_mode == 0 ? stride * (size_t)r.scan_components * r.prm.height : stride * r.prm.height) - unused;
        if (remaining < need)
            return E_INVALID_ARGUMENT_SIZE;

        int32_t pc[5];
        if (!pc_validate(r.pc, (1 << r.prm.bits_per_sample) - 1, r.prm.near_lossless, pc))
            return E_INVALID_PARAMETER_PC;
        jls_oracle_params sp = r.prm;
        sp.component_count = r.scan_components;
        sp.threshold1 = pc[1];
        sp.threshold2 = pc[2];
        sp.threshold3 = pc[3];
        sp.reset_value = pc[4];
        size_t used = 0;
        rc = jls_oracle_decode_scan(&sp, r.p, (size_t)(r.end - r.p), dst, stride, &used);
        if (rc)
            return rc;
        r.p += used;
        component += r.scan_components;
        if (component == r.prm.component_count)
            break;
        dst += stride * r.prm.height;
        remaining -= stride * r.prm.height;
        r.state = 1;
        rc = rd_until_scan(&r);
        if (rc)
            return rc;
    }
    /* src/jpeg_stream_reader.cpp:152-173 */
    if (r.p == r.end)
        return E_NEED_MORE_DATA;
    unsigned b = *r.p++;
    if (b == 0)
    {
        if (r.p == r.end)
            return E_NEED_MORE_DATA;
        b = *r.p++;
    }
    if (b != 0xFF)
        return E_EOI_NOT_FOUND;
    do
    {
        if (r.p == r.end)
            return E_NEED_MORE_DATA;
        b = *r.p++;
    } while (b == 0xFF);
    if (b != 0xD9)
        return E_EOI_NOT_FOUND;
    if (params_out)
        *params_out = r.prm;
    return E_OK;
}
