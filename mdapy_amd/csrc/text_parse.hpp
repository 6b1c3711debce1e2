// text_parse.hpp — decimal text -> IEEE double / int32, written once for host and device.
//
// Input side of the hot path (reference: src/mdapy/load_save.py:66-198 dump frames, :653-863 extended XYZ —
// both hand the atom table to a CSV tokenizer and end up with what Python's float() returns for every field).  The
// conversion here is the Eisel-Lemire algorithm (Lemire, "Number parsing at a gigabyte per second", SPE 2021; Mushtak &
// Lemire 2023 prove that with a 128-bit table of powers of five no fallback is needed for a 64-bit decimal significand):
// correctly rounded, i.e. bit-identical to float(), for every field of at most 19 significant digits.  Longer fields
// are converted twice (truncated significand w and w + 1); only when the two disagree is the field reported back to the
// caller for re-parsing on the host.  The table (651 x 128 bit) is generated exactly with integer arithmetic by
// text_pow5_table() — not transcribed.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define TXT_HD __host__ __device__ __forceinline__
#else
#define TXT_HD inline
#endif

namespace mdtext {

constexpr int POW5_MIN = -342, POW5_MAX = 308, POW5_N = POW5_MAX - POW5_MIN + 1;

TXT_HD void mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo)
{
#ifdef __HIP_DEVICE_COMPILE__
    *lo = a * b;
    *hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p;
    *hi = (uint64_t)(p >> 64);
#endif
}

TXT_HD int clz64(uint64_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

// bits of the double nearest to  w * 10^q  (w != 0 handled, sign excluded); pow5[2 (q + 342)] = high, +1 = low word
TXT_HD uint64_t decimal_to_bits(uint64_t w, int64_t q, const uint64_t *pow5)
{
    if (w == 0 || q < POW5_MIN)
        return 0;
    if (q > POW5_MAX)
        return 0x7FF0000000000000ull;
    const int lz = clz64(w);
    w <<= lz;
    const uint64_t *t = pow5 + 2 * (q - POW5_MIN);
    uint64_t hi, lo;
    mul64(w, t[0], &hi, &lo);
    if ((hi & 0x1FF) == 0x1FF) { // the 55 leading bits are not settled by the high word of the power alone
        uint64_t hi2, lo2;
        mul64(w, t[1], &hi2, &lo2);
        lo += hi2;
        if (hi2 > lo)
            ++hi;
    }
    const int upper = (int)(hi >> 63);
    uint64_t m = hi >> (upper + 9);
    int64_t e2 = (((int64_t)(152170 + 65536) * q) >> 16) + 63 + upper - lz + 1023;
    if (e2 <= 0) { // subnormal
        if (-e2 + 1 >= 64)
            return 0;
        m >>= -e2 + 1;
        m += m & 1;
        m >>= 1;
        return m; // exponent field 0 (or 1 when the rounding carried into bit 52: the same bit pattern)
    }
    if (lo <= 1 && q >= -4 && q <= 23 && (m & 3) == 1 && (m << (upper + 9)) == hi)
        m &= ~(uint64_t)1; // exactly halfway: round to even
    m += m & 1;
    m >>= 1;
    if (m >= ((uint64_t)2 << 52)) {
        m = (uint64_t)1 << 52;
        ++e2;
    }
    m &= ~((uint64_t)1 << 52);
    if (e2 >= 0x7FF)
        return 0x7FF0000000000000ull;
    return m | ((uint64_t)e2 << 52);
}

enum { TOK_OK = 0, TOK_REDO = 1, TOK_BAD = 2 };

// one floating-point field [p, end): optional sign, digits, optional fraction, optional exponent.  TOK_REDO: not
// decidable here (more than 19 significant digits with a rounding boundary in between, nan / inf spellings, hex floats):
// the caller re-parses it on the host.  TOK_BAD: not a number at all.
TXT_HD int parse_double(const char *p, const char *end, const uint64_t *pow5, double *out)
{
    const char *s = p;
    bool neg = false;
    if (s < end && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    uint64_t w = 0;
    int nd = 0;          // significant digits taken into w (leading zeros do not count)
    int64_t dropped = 0; // integer-part digits beyond the 19th
    bool more = false;   // a non-zero digit was dropped
    bool any = false;
    int64_t q = 0;
    for (; s < end && *s >= '0' && *s <= '9'; ++s) {
        any = true;
        const int d = *s - '0';
        if (nd < 19) { if (w != 0 || d != 0) { w = w * 10 + d; ++nd; } }
        else { ++dropped; more = more || d != 0; }
    }
    if (s < end && *s == '.') {
        ++s;
        for (; s < end && *s >= '0' && *s <= '9'; ++s) {
            any = true;
            const int d = *s - '0';
            if (nd < 19) { if (w != 0 || d != 0) { w = w * 10 + d; ++nd; } --q; }
            else { more = more || d != 0; }
        }
    }
    if (!any)
        return (s < end && ((*s | 32) == 'n' || (*s | 32) == 'i')) ? TOK_REDO : TOK_BAD;
    q += dropped;
    if (s < end && (*s | 32) == 'e') {
        ++s;
        bool eneg = false;
        if (s < end && (*s == '-' || *s == '+')) { eneg = *s == '-'; ++s; }
        if (!(s < end && *s >= '0' && *s <= '9'))
            return TOK_BAD;
        int64_t ex = 0;
        for (; s < end && *s >= '0' && *s <= '9'; ++s)
            if (ex < 100000) ex = ex * 10 + (*s - '0');
        q += eneg ? -ex : ex;
    }
    if (s != end)
        return (*s == 'x' || *s == 'X') ? TOK_REDO : TOK_BAD;
    uint64_t bits = decimal_to_bits(w, q, pow5);
    if (more) { // the true significand lies strictly between w and w + 1
        const uint64_t up = decimal_to_bits(w + 1, q, pow5);
        if (up != bits)
            return TOK_REDO;
    }
    if (neg) bits |= 0x8000000000000000ull;
    union { uint64_t u; double d; } cv;
    cv.u = bits;
    *out = cv.d;
    return TOK_OK;
}

TXT_HD int parse_int32(const char *p, const char *end, int *out)
{
    const char *s = p;
    bool neg = false;
    if (s < end && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    if (s == end)
        return TOK_BAD;
    int64_t v = 0;
    for (; s < end; ++s) {
        if (*s < '0' || *s > '9')
            return TOK_REDO; // "3.0", "1e3": let the host decide what the reader accepts
        v = v * 10 + (*s - '0');
        if (v > 4294967296ll)
            return TOK_REDO;
    }
    v = neg ? -v : v;
    if (v < -2147483648ll || v > 2147483647ll)
        return TOK_REDO;
    *out = (int)v;
    return TOK_OK;
}

} // namespace mdtext
