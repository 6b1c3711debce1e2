// text.hip — the atom table of a LAMMPS dump / extended-XYZ file, tokenised and converted in HBM.
//
// Reference behaviour: src/mdapy/load_save.py:66-198 (dump frame: 9 header lines, then one row per atom, columns named by
// "ITEM: ATOMS ...", integer columns id/type/ix/..., string columns element/typelabel, everything else float64) and
// :653-863 (extended XYZ: Properties= gives name:type:count triples).  The reference slurps the file and calls a CSV
// reader; here the host only streams the bytes into HBM (pinned, chunked, overlapping the read — load_save.py of this
// package) and three kernels do the rest:
//   k_count_lines   newline count per 4 KB block                       (reads the text once, 16 B per lane)
//   k_line_starts   row r -> byte offset of its first character        (prefix of the counts + a block scan)
//   k_parse_rows    one row per lane: split at blanks, convert field c to f64 / i32 / 8 packed characters, store
//                   column-major (coalesced 8 / 4 byte stores).  Conversion = text_parse.hpp (correctly rounded).
// Everything is byte / integer work bound by HBM: a 500 MB dump is read twice and ~0.3 GB of columns written.
// Fields the device cannot decide (more than 19 significant digits on a rounding boundary, nan/inf, > 8 character
// strings) are listed in `redo` and re-parsed by the caller with the host's float(): results are what the reference gets.
#include "common.hpp"
#include "grid.hpp"
#include "text_parse.hpp"
#include <mutex>
#include <vector>

namespace mdh {

// ---------------------------------------------------------------------------------------------------------------------
// 128-bit truncated powers of five, 5^q for q in [-342, 308], generated with exact integer arithmetic
// (normalised so that bit 127 is set; negative powers are 2^b / 5^-q rounded up — Lemire 2021, section 5)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Big { // little-endian base 2^32
    std::vector<uint32_t> w;
    void trim() { while (!w.empty() && w.back() == 0) w.pop_back(); }
    int bits() const { return w.empty() ? 0 : (int)(32 * (w.size() - 1) + 32 - __builtin_clz(w.back())); }
    void mul_small(uint32_t m)
    {
        uint64_t c = 0;
        for (auto &x : w) { c += (uint64_t)x * m; x = (uint32_t)c; c >>= 32; }
        if (c) w.push_back((uint32_t)c);
    }
    void shl1(uint32_t in)
    {
        uint32_t c = in;
        for (auto &x : w) { const uint32_t n = x >> 31; x = (x << 1) | c; c = n; }
        if (c) w.push_back(c);
    }
    bool ge(const Big &o) const
    {
        if (w.size() != o.w.size()) return w.size() > o.w.size();
        for (size_t i = w.size(); i-- > 0;)
            if (w[i] != o.w[i]) return w[i] > o.w[i];
        return true;
    }
    void sub(const Big &o)
    {
        int64_t c = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            int64_t v = (int64_t)w[i] - (i < o.w.size() ? o.w[i] : 0) + c;
            c = v < 0 ? -1 : 0;
            w[i] = (uint32_t)(v & 0xFFFFFFFFll);
        }
        trim();
    }
    bool bit(int i) const { return (size_t)(i / 32) < w.size() && ((w[i / 32] >> (i % 32)) & 1u); }
    void top128(uint64_t *hi, uint64_t *lo) const // the 128 leading bits (value must have >= 128 bits)
    {
        const int n = bits();
        uint64_t h = 0, l = 0;
        for (int k = 0; k < 64; ++k) h = (h << 1) | (bit(n - 1 - k) ? 1u : 0u);
        for (int k = 64; k < 128; ++k) l = (l << 1) | (bit(n - 1 - k) ? 1u : 0u);
        *hi = h; *lo = l;
    }
};
} // namespace

void text_pow5_table(uint64_t *out) // [2 * POW5_N]: high, low
{
    using namespace mdtext;
    for (int q = 0; q <= POW5_MAX; ++q) {
        Big p; p.w = {1};
        for (int i = 0; i < q; ++i) p.mul_small(5);
        while (p.bits() < 128) p.shl1(0);
        p.top128(&out[2 * (q - POW5_MIN)], &out[2 * (q - POW5_MIN) + 1]);
    }
    for (int q = POW5_MIN; q < 0; ++q) {
        Big p; p.w = {1};
        for (int i = 0; i < -q; ++i) p.mul_small(5);
        int z = 0; // smallest z with 2^z >= 5^-q
        { Big one; one.w = {1}; while (!one.ge(p)) { one.shl1(0); ++z; } }
        const int b = q >= -27 ? z + 127 : 2 * z + 128;
        // c = floor(2^b / p) + 1 by binary long division
        Big rem, quo;
        quo.w.assign((size_t)(b / 32 + 1), 0);
        for (int i = b; i >= 0; --i) {
            rem.shl1(i == b ? 1u : 0u);
            if (rem.w.empty() && i != b) { /* still zero */ }
            if (rem.ge(p) && !rem.w.empty()) { rem.sub(p); quo.w[i / 32] |= 1u << (i % 32); }
        }
        quo.trim();
        { // + 1
            size_t i = 0;
            while (true) {
                if (i == quo.w.size()) { quo.w.push_back(1); break; }
                if (++quo.w[i] != 0) break;
                ++i;
            }
        }
        quo.top128(&out[2 * (q - POW5_MIN)], &out[2 * (q - POW5_MIN) + 1]); // "truncate while c >= 2^128"
    }
}

static const uint64_t *host_pow5()
{
    static std::once_flag once;
    static uint64_t table[2 * mdtext::POW5_N];
    std::call_once(once, [] { text_pow5_table(table); });
    return table;
}

static int device_pow5(const uint64_t **out)
{
    static std::mutex mu;
    static const uint64_t *dev[64] = {nullptr};
    int d = 0;
    MDH_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lk(mu);
    if (d < 0 || d >= 64) { set_error("mdh_parse_table: device ordinal out of range"); return MDH_ERR_HIP; }
    if (!dev[d]) {
        void *p = nullptr;
        MDH_HIP(hipMalloc(&p, sizeof(uint64_t) * 2 * mdtext::POW5_N));
        MDH_HIP(hipMemcpy(p, host_pow5(), sizeof(uint64_t) * 2 * mdtext::POW5_N, hipMemcpyHostToDevice));
        dev[d] = static_cast<const uint64_t *>(p);
    }
    *out = dev[d];
    return MDH_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TXT_THREADS = 256, TXT_PER_LANE = 16, TXT_BLOCK_BYTES = TXT_THREADS * TXT_PER_LANE;

__device__ __forceinline__ unsigned newline_count16(const char *__restrict__ text, int64_t at, int64_t nbytes)
{
    unsigned c = 0;
    if (at + TXT_PER_LANE <= nbytes && ((reinterpret_cast<uintptr_t>(text + at) & 15) == 0)) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + at); // one 16-byte load per lane, 1 KB per wavefront
        const unsigned wds[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k)
            for (int s = 0; s < 32; s += 8) c += ((wds[k] >> s) & 0xFFu) == 0x0Au ? 1u : 0u;
    } else {
        for (int k = 0; k < TXT_PER_LANE; ++k)
            if (at + k < nbytes && text[at + k] == '\n') ++c;
    }
    return c;
}

__global__ __launch_bounds__(TXT_THREADS) void k_count_lines(const char *__restrict__ text, int64_t nbytes, unsigned *__restrict__ counts)
{
    __shared__ unsigned part[TXT_THREADS / 64];
    const int64_t at = ((int64_t)blockIdx.x * TXT_THREADS + threadIdx.x) * TXT_PER_LANE;
    unsigned c = at < nbytes ? newline_count16(text, at, nbytes) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// starts[r] = offset of the first byte of row r (row 0 starts at 0; row r > 0 one past the r-th newline), r < nrows + 1
__global__ __launch_bounds__(TXT_THREADS) void k_line_starts(const char *__restrict__ text, int64_t nbytes, const int *__restrict__ prefix,
                                                             int64_t nrows, int64_t *__restrict__ starts)
{
    __shared__ unsigned wsum[TXT_THREADS / 64];
    const int64_t at = ((int64_t)blockIdx.x * TXT_THREADS + threadIdx.x) * TXT_PER_LANE;
    const unsigned c = at < nbytes ? newline_count16(text, at, nbytes) : 0u;
    // exclusive scan of c over the block
    unsigned inc = c;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    int64_t row = (int64_t)prefix[blockIdx.x] + base + (inc - c) + 1; // index of the row that begins after this lane's first newline
    if (blockIdx.x == 0 && threadIdx.x == 0) starts[0] = 0;
    if (c) {
        for (int k = 0; k < TXT_PER_LANE; ++k)
            if (at + k < nbytes && text[at + k] == '\n') {
                if (row <= nrows) starts[row] = at + k + 1;
                ++row;
            }
    }
}

struct ParseOut {
    void *col[64];
    int kind[64];
};

__device__ __forceinline__ bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

__global__ __launch_bounds__(256) void k_parse_rows(const char *__restrict__ text, int64_t nbytes, const int64_t *__restrict__ starts,
                                                    int64_t nrows, int64_t nlines, int64_t nnewlines, int ncol, ParseOut out, const uint64_t *__restrict__ pow5,
                                                    int64_t *__restrict__ redo, int64_t redo_cap, unsigned long long *__restrict__ status)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows)
        return;
    if (r >= nlines) { // the text ends before this row
        atomicAdd(&status[1], 1ull);
        return;
    }
    const char *p = text + starts[r];
    const char *end = r < nnewlines ? text + starts[r + 1] - 1 : text + nbytes; // the row's newline, or the end of a text without a last one
    int c = 0;
    while (c < ncol) {
        while (p < end && is_blank(*p)) ++p;
        if (p >= end)
            break;
        const char *q = p;
        while (q < end && !is_blank(*q)) ++q;
        const int kind = out.kind[c];
        int st = mdtext::TOK_OK;
        if (kind == 0) {
            double v = 0;
            st = mdtext::parse_double(p, q, pow5, &v);
            static_cast<double *>(out.col[c])[r] = v;
        } else if (kind == 1) {
            int v = 0;
            st = mdtext::parse_int32(p, q, &v);
            static_cast<int *>(out.col[c])[r] = v;
        } else if (kind == 2) {
            unsigned long long v = 0;
            if (q - p > 8) st = mdtext::TOK_REDO;
            else for (int k = 0; k < (int)(q - p); ++k) v |= (unsigned long long)(unsigned char)p[k] << (8 * k);
            static_cast<unsigned long long *>(out.col[c])[r] = v;
        }
        if (st != mdtext::TOK_OK) {
            const unsigned long long slot = atomicAdd(&status[0], 1ull);
            if ((int64_t)slot < redo_cap) {
                redo[2 * slot] = r * ncol + c;
                redo[2 * slot + 1] = ((int64_t)(p - text) << 16) | (int64_t)(q - p > 65535 ? 65535 : q - p);
            }
            if (st == mdtext::TOK_BAD) atomicAdd(&status[2], 1ull);
        }
        p = q;
        ++c;
    }
    if (c < ncol)
        atomicAdd(&status[1], 1ull); // a short row
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_debug_text_pow5(int q, uint64_t *out2)
{
    if (q < mdtext::POW5_MIN || q > mdtext::POW5_MAX)
        return MDH_ERR_ARG;
    const uint64_t *t = host_pow5() + 2 * (q - mdtext::POW5_MIN);
    out2[0] = t[0];
    out2[1] = t[1];
    return MDH_OK;
}

// host twin of the field converter (tests pin it against float() without a GPU): 0 ok, 1 undecided, 2 not a number
extern "C" int mdh_debug_parse_double(const char *s, int64_t len, double *out)
{
    return mdtext::parse_double(s, s + len, host_pow5(), out);
}

extern "C" int mdh_parse_table(const char *text, int64_t nbytes, int text_space, int64_t nrows, int ncol, const int *kinds,
                               void *const *columns, int64_t *redo, int64_t redo_cap, int64_t *status4, int space, void *stream)
{
    if (nbytes < 0 || nrows < 0 || ncol < 1 || ncol > 64 || redo_cap < 0) {
        set_error("mdh_parse_table: 1..64 columns, non-negative sizes");
        return MDH_ERR_ARG;
    }
    for (int c = 0; c < ncol; ++c)
        if (kinds[c] < 0 || kinds[c] > 3) { set_error("mdh_parse_table: column kind must be 0 (f64), 1 (i32), 2 (8 characters) or 3 (skip)"); return MDH_ERR_ARG; }
    if (nbytes > 2147483647LL * mdh::TXT_BLOCK_BYTES / 2) { set_error("mdh_parse_table: text too large for one call"); return MDH_ERR_ARG; }
    Scope sc(stream);
    if (sc.failed())
        return sc.error();
    const uint64_t *dpow = nullptr;
    MDH_TRY(device_pow5(&dpow));
    hipStream_t st = sc.stream();
    const char *dtext = sc.stage_in(text, (size_t)nbytes, text_space);
    ParseOut po;
    size_t width[4] = {8, 4, 8, 0};
    for (int c = 0; c < 64; ++c) { po.col[c] = nullptr; po.kind[c] = 3; }
    for (int c = 0; c < ncol; ++c) {
        po.kind[c] = kinds[c];
        if (kinds[c] != 3)
            po.col[c] = sc.stage(static_cast<unsigned char *>(columns[c]), (size_t)nrows * width[kinds[c]], space, false, true);
    }
    int64_t *dredo = sc.stage(redo, (size_t)redo_cap * 2, space, false, true);
    unsigned long long *dstatus = sc.alloc_n<unsigned long long>(4);
    const int64_t nblk = (nbytes + TXT_BLOCK_BYTES - 1) / TXT_BLOCK_BYTES;
    unsigned *counts = sc.alloc_n<unsigned>((size_t)nblk + 1);
    int *prefix = sc.alloc_n<int>((size_t)nblk + 2);
    int64_t *starts = sc.alloc_n<int64_t>((size_t)nrows + 2);
    if (sc.failed())
        return sc.error();
    MDH_HIP(hipMemsetAsync(dstatus, 0, 4 * sizeof(unsigned long long), st));
    int64_t nlines = 0;
    if (nbytes > 0 && nrows > 0) {
        {
            ProfRange pr("k_count_lines", st);
            hipLaunchKernelGGL(k_count_lines, dim3((unsigned)nblk), dim3(TXT_THREADS), 0, st, dtext, nbytes, counts);
        }
        MDH_TRY(exclusive_scan_u32(sc, counts, prefix, nblk));
        {
            ProfRange pr("k_line_starts", st);
            hipLaunchKernelGGL(k_line_starts, dim3((unsigned)nblk), dim3(TXT_THREADS), 0, st, dtext, nbytes, prefix, nrows, starts);
        }
        // rows present = newlines (+ 1 when the text does not end with one): needed on the host to size the parse
        int total = 0;
        char last = 0;
        MDH_HIP(hipMemcpyAsync(&total, prefix + nblk, sizeof(int), hipMemcpyDeviceToHost, st));
        MDH_HIP(hipMemcpyAsync(&last, dtext + nbytes - 1, 1, hipMemcpyDeviceToHost, st));
        MDH_HIP(hipStreamSynchronize(st));
        nlines = (int64_t)total + (last == '\n' ? 0 : 1);
        ProfRange pr("k_parse_rows", st);
        hipLaunchKernelGGL(k_parse_rows, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, dtext, nbytes, starts, nrows, nlines, (int64_t)total, ncol, po, dpow,
                           dredo, redo_cap, dstatus);
    }
    unsigned long long hs[4] = {0, 0, 0, 0};
    MDH_HIP(hipMemcpyAsync(hs, dstatus, sizeof(hs), hipMemcpyDeviceToHost, st));
    MDH_HIP(hipStreamSynchronize(st));
    status4[0] = (int64_t)hs[0]; // fields to re-parse on the host (listed in redo up to redo_cap)
    status4[1] = (int64_t)hs[1]; // rows missing or shorter than ncol fields
    status4[2] = (int64_t)hs[2]; // fields that are not numbers at all
    status4[3] = nlines;
    return sc.finish(space);
}

MDH_WARM_UNIT(text)
