// libvodhip -- the wire codec of the search service (HOST code only; compiled as plain C++, see the Makefile):
// urlsafe base64 of `head || data` and back, byte-identical to Python's base64.urlsafe_b64encode / urlsafe_b64decode
// (/root/reference/src/vod_search/io.py:17-32).  Declared in include/vodhip.h.
//
// Three implementations behind one entry point each, chosen once per process (cpuid): AVX-512 VBMI (below), and
//   * AVX2 (Mula / Lemire style): 24 input bytes -> 32 characters per iteration (two multiplies split the 6-bit fields, one byte shuffle
//     maps them to the alphabet); 32 characters -> 24 bytes (nibble look-ups translate and validate, two multiply-adds pack).
//     Round 4: a 1024 x 768 float32 query batch is 4.2 MB of base64 - at the table-driven loops' ~2 GB/s that was ~2 ms per direction
//     and per side, most of what /fast-search (the reference's wire format) cost over the raw-bytes route.
//   * table-driven scalar loops (round 3): 12 input bits -> two characters per look-up; four pre-shifted decode tables.  Also the
//     head / seam / tail handling of the AVX2 path and the fallback for characters outside the urlsafe alphabet ('+', '/').
#include <stdint.h>
#include <string.h>

#include <immintrin.h>

#include "../../include/vodhip.h"

namespace {

const char kB64Url[65] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_";

struct B64Tables {
    uint16_t enc12[4096];  // 12 bits -> two characters (little endian: first character in the low byte)
    uint32_t dec[4][256];  // character at position i of a quantum -> its 6 bits, shifted into place; bit 31 = invalid
    B64Tables() {
        for (int v = 0; v < 4096; ++v) enc12[v] = (uint16_t)((unsigned char)kB64Url[v >> 6] | ((unsigned char)kB64Url[v & 63] << 8));
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 256; ++c) dec[i][c] = 0x80000000u;
        for (uint32_t v = 0; v < 64; ++v) put((unsigned char)kB64Url[v], v);
        put('+', 62);
        put('/', 63);
    }
    void put(unsigned char c, uint32_t v) {
        for (int i = 0; i < 4; ++i) dec[i][c] = v << (18 - 6 * i);
    }
};
const B64Tables& tables() {
    static const B64Tables t;  // thread-safe initialisation (C++11 magic static)
    return t;
}

bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
bool have_vbmi() {
    static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi");
    return yes;
}

// ---- AVX-512 VBMI (Zen 4 / Zen 5, Ice Lake+): one byte permute gathers the triples, one multishift extracts the four 6-bit fields, one
// byte permute looks the characters up: 48 bytes -> 64 characters; decoding is a 128-entry two-table permute + the two multiply-adds.
#define VBMI __attribute__((target("avx512f,avx512bw,avx512vbmi")))
VBMI int64_t encode_vbmi(const uint8_t* p, int64_t n, char* o) {
    alignas(64) uint8_t gather[64];
    for (int i = 0; i < 16; ++i) {
        gather[4 * i + 0] = (uint8_t)(3 * i + 1);
        gather[4 * i + 1] = (uint8_t)(3 * i + 0);
        gather[4 * i + 2] = (uint8_t)(3 * i + 2);
        gather[4 * i + 3] = (uint8_t)(3 * i + 1);
    }
    const __m512i idx = _mm512_load_si512(gather);
    const __m512i shifts = _mm512_set1_epi64(0x3036242a1016040aLL);
    const __m512i lut = _mm512_loadu_si512(kB64Url);
    int64_t i = 0;
    for (; i + 64 <= n; i += 48, o += 64) {  // (reads 64 bytes, consumes 48)
        const __m512i in = _mm512_permutexvar_epi8(idx, _mm512_loadu_si512(p + i));
        const __m512i six = _mm512_multishift_epi64_epi8(shifts, in);
        _mm512_storeu_si512(o, _mm512_permutexvar_epi8(six, lut));  // (vpermb looks at the low 6 bits of each index only)
    }
    return i;
}

// both alphabets at once ('+' = '-' = 62, '/' = '_' = 63); anything else (padding, whitespace, junk, bytes >= 0x80) stops the vector loop
VBMI int64_t decode_vbmi(const unsigned char* s, int64_t n, uint8_t* o) {
    alignas(64) uint8_t table[128];
    memset(table, 0x80, sizeof(table));
    for (int v = 0; v < 64; ++v) table[(unsigned char)kB64Url[v]] = (uint8_t)v;
    table[(unsigned char)'+'] = 62;
    table[(unsigned char)'/'] = 63;
    const __m512i lut_lo = _mm512_load_si512(table), lut_hi = _mm512_load_si512(table + 64);
    alignas(64) uint8_t pack[64];
    memset(pack, 0, sizeof(pack));
    for (int j = 0; j < 16; ++j) {
        pack[3 * j + 0] = (uint8_t)(4 * j + 2);
        pack[3 * j + 1] = (uint8_t)(4 * j + 1);
        pack[3 * j + 2] = (uint8_t)(4 * j + 0);
    }
    const __m512i pack_idx = _mm512_load_si512(pack);
    int64_t i = 0;
    for (; i + 64 <= n; i += 64, o += 48) {
        const __m512i src = _mm512_loadu_si512(s + i);
        const __m512i v = _mm512_permutex2var_epi8(lut_lo, src, lut_hi);  // 7-bit index; bit 7 of a byte >= 0x80 is checked below
        if (_mm512_movepi8_mask(_mm512_or_si512(v, src)) != 0) break;
        const __m512i ab = _mm512_maddubs_epi16(v, _mm512_set1_epi32(0x01400140));
        const __m512i packed = _mm512_madd_epi16(ab, _mm512_set1_epi32(0x00011000));
        _mm512_mask_storeu_epi8(o, 0x0000FFFFFFFFFFFFull, _mm512_permutexvar_epi8(pack_idx, packed));
    }
    return i;
}

// ---- AVX2 -------------------------------------------------------------------------------------------------------------------------
// encode whole 24-byte groups of p[0, n): returns the number of input bytes consumed (a multiple of 24); reads up to 4 bytes past
// each group's 24, so it stops while at least 28 bytes remain
__attribute__((target("avx2"))) int64_t encode_avx2(const uint8_t* p, int64_t n, char* o) {
    const __m256i shuf = _mm256_setr_epi8(1, 0, 2, 1, 4, 3, 5, 4, 7, 6, 8, 7, 10, 9, 11, 10, 1, 0, 2, 1, 4, 3, 5, 4, 7, 6, 8, 7, 10, 9, 11, 10);
    // offset of each class of 6-bit value to its character: 26..51 -> 'a' - 26, 52..61 -> '0' - 52, 62 -> '-' - 62, 63 -> '_' - 63, 0..25 -> 'A'
    const __m256i lut = _mm256_setr_epi8(71, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -17, 32, 65, 0, 0, 71, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -17, 32,
                                         65, 0, 0);
    int64_t i = 0;
    for (; i + 28 <= n; i += 24, o += 32) {
        const __m128i lo = _mm_loadu_si128((const __m128i*)(p + i));
        const __m128i hi = _mm_loadu_si128((const __m128i*)(p + i + 12));
        __m256i in = _mm256_shuffle_epi8(_mm256_set_m128i(hi, lo), shuf);
        const __m256i t0 = _mm256_and_si256(in, _mm256_set1_epi32(0x0fc0fc00));
        const __m256i t1 = _mm256_mulhi_epu16(t0, _mm256_set1_epi32(0x04000040));
        const __m256i t2 = _mm256_and_si256(in, _mm256_set1_epi32(0x003f03f0));
        const __m256i t3 = _mm256_mullo_epi16(t2, _mm256_set1_epi32(0x01000010));
        const __m256i idx = _mm256_or_si256(t1, t3);  // one 6-bit value per byte
        __m256i red = _mm256_subs_epu8(idx, _mm256_set1_epi8(51));
        const __m256i less = _mm256_cmpgt_epi8(_mm256_set1_epi8(26), idx);
        red = _mm256_or_si256(red, _mm256_and_si256(less, _mm256_set1_epi8(13)));
        _mm256_storeu_si256((__m256i*)o, _mm256_add_epi8(idx, _mm256_shuffle_epi8(lut, red)));
    }
    return i;
}

// decode whole 32-character groups of s[0, n) that lie entirely in the urlsafe alphabet: returns the number of characters consumed
// (a multiple of 32; stops at the first group with any other character, which the scalar loop then judges); writes 24 bytes per group
__attribute__((target("avx2"))) int64_t decode_avx2(const unsigned char* s, int64_t n, uint8_t* o) {
    // per high nibble: the bit that stands for it (0 = no character of the alphabet has it) and the offset character -> value
    const __m256i lut_hi = _mm256_setr_epi8(0, 0, 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0, 0,
                                            0, 0, 0, 0, 0, 0);
    // per low nibble: the high nibbles it may appear with ('-' 2D; '0'-'9' 30-39; 'A'-'O' 41-4F; 'P'-'Z' 50-5A, '_' 5F; 'a'-'o' 61-6F; 'p'-'z' 70-7A)
    const __m256i lut_lo = _mm256_setr_epi8(0x2A, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3C, 0x14, 0x14, 0x15, 0x14, 0x1C, 0x2A, 0x3E, 0x3E,
                                            0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3E, 0x3C, 0x14, 0x14, 0x15, 0x14, 0x1C);
    const __m256i lut_roll = _mm256_setr_epi8(0, 0, 17, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 17, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i pack_shuf = _mm256_setr_epi8(2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1, 2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1);
    int64_t i = 0;
    for (; i + 32 <= n; i += 32, o += 24) {
        const __m256i src = _mm256_loadu_si256((const __m256i*)(s + i));
        const __m256i hi_n = _mm256_and_si256(_mm256_srli_epi32(src, 4), _mm256_set1_epi8(0x0f));
        const __m256i lo_n = _mm256_and_si256(src, _mm256_set1_epi8(0x0f));
        const __m256i ok = _mm256_and_si256(_mm256_shuffle_epi8(lut_hi, hi_n), _mm256_shuffle_epi8(lut_lo, lo_n));
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(ok, _mm256_setzero_si256())) != 0) break;  // '+', '/', '=', whitespace, junk: scalar path
        const __m256i underscore = _mm256_cmpeq_epi8(src, _mm256_set1_epi8(0x5f));            // '_' shares its high nibble with 'P'-'Z'
        const __m256i roll = _mm256_add_epi8(_mm256_shuffle_epi8(lut_roll, hi_n), _mm256_and_si256(underscore, _mm256_set1_epi8(33)));
        const __m256i v = _mm256_add_epi8(src, roll);
        const __m256i ab = _mm256_maddubs_epi16(v, _mm256_set1_epi32(0x01400140));
        const __m256i packed = _mm256_madd_epi16(ab, _mm256_set1_epi32(0x00011000));
        const __m256i bytes = _mm256_shuffle_epi8(packed, pack_shuf);  // 12 bytes at the bottom of each 128-bit lane
        _mm_storeu_si128((__m128i*)o, _mm256_castsi256_si128(bytes));              // 12 + 4 scratch bytes (the next store overwrites them)
        const __m128i up = _mm256_extracti128_si256(bytes, 1);
        _mm_storel_epi64((__m128i*)(o + 12), up);                                   // bytes 12..19
        const uint32_t last = (uint32_t)_mm_extract_epi32(up, 2);                   // bytes 20..23
        memcpy(o + 20, &last, 4);
    }
    return i;
}

}  // namespace

extern "C" {

int64_t vodhip_b64url_encode(const uint8_t* head, int64_t n_head, const uint8_t* data, int64_t n_data, char* out) {
    if (n_head < 0 || n_data < 0 || !out || (n_head && !head) || (n_data && !data)) return -1;
    const B64Tables& T = tables();
    const int64_t n = n_head + n_data;
    auto at = [&](int64_t i) -> uint32_t { return i < n_head ? head[i] : data[i - n_head]; };
    char* o = out;
    int64_t i = 0;
    // the head and the triple that straddles the head / data seam go through the generic accessor
    const int64_t seam_end = n_head == 0 ? 0 : ((n_head + 2) / 3) * 3;
    for (; i + 2 < n && i < seam_end; i += 3) {
        const uint32_t v = (at(i) << 16) | (at(i + 1) << 8) | at(i + 2);
        const uint16_t a = T.enc12[v >> 12], b = T.enc12[v & 4095];
        memcpy(o, &a, 2);
        memcpy(o + 2, &b, 2);
        o += 4;
    }
    if (i + 2 < n) {
        const uint8_t* p = data + (i - n_head);  // i >= n_head here
        if (have_vbmi()) {
            const int64_t done = encode_vbmi(p, n - i, o);
            i += done;
            p += done;
            o += done / 3 * 4;
        }
        if (have_avx2()) {
            const int64_t done = encode_avx2(p, n - i, o);
            i += done;
            p += done;
            o += done / 3 * 4;
        }
        for (; i + 5 < n; i += 6, p += 6, o += 8) {  // two quanta per iteration
            const uint32_t v0 = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
            const uint32_t v1 = ((uint32_t)p[3] << 16) | ((uint32_t)p[4] << 8) | p[5];
            const uint64_t w = (uint64_t)T.enc12[v0 >> 12] | ((uint64_t)T.enc12[v0 & 4095] << 16) | ((uint64_t)T.enc12[v1 >> 12] << 32) |
                               ((uint64_t)T.enc12[v1 & 4095] << 48);
            memcpy(o, &w, 8);
        }
        for (; i + 2 < n; i += 3, p += 3, o += 4) {
            const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
            const uint32_t w = (uint32_t)T.enc12[v >> 12] | ((uint32_t)T.enc12[v & 4095] << 16);
            memcpy(o, &w, 4);
        }
    }
    if (i < n) {
        const uint32_t b0 = at(i), b1 = i + 1 < n ? at(i + 1) : 0;
        *o++ = kB64Url[b0 >> 2];
        *o++ = kB64Url[((b0 & 3) << 4) | (b1 >> 4)];
        *o++ = i + 1 < n ? kB64Url[(b1 & 15) << 2] : '=';
        *o++ = '=';
    }
    return (int64_t)(o - out);
}

int64_t vodhip_b64url_decode(const char* src, int64_t n, uint8_t* out) {
    if (n < 0 || (n && (!src || !out))) return -1;
    const B64Tables& T = tables();
    const unsigned char* s = (const unsigned char*)src;
    while (n > 0 && s[n - 1] == '=') --n;
    uint8_t* o = out;
    int64_t i = 0;
    if (have_vbmi() && n >= 128) {
        i = decode_vbmi(s, n - 64, o);
        o += i / 4 * 3;
    }
    if (have_avx2() && n - i >= 64) {
        // (the vector loops write exactly 3 bytes per 4 characters; they leave the last group to the scalar loops, whose 4-byte store of a
        // quantum needs one more quantum behind it)
        const int64_t done = decode_avx2(s + i, n - i - 32, o);
        i += done;
        o += done / 4 * 3;
    }
    uint32_t bad = 0;
    // the 4-byte store of a quantum spills one byte past its 3: safe while at least one more quantum (or the tail) follows
    for (; i + 7 < n; i += 4, o += 3) {
        const uint32_t v = T.dec[0][s[i]] | T.dec[1][s[i + 1]] | T.dec[2][s[i + 2]] | T.dec[3][s[i + 3]];
        bad |= v;
        const uint32_t be = __builtin_bswap32(v << 8);  // bytes v[23:16], v[15:8], v[7:0], 0 in memory order
        memcpy(o, &be, 4);
    }
    if (bad & 0x80000000u) return -1;
    for (; i + 3 < n; i += 4, o += 3) {
        const uint32_t v = T.dec[0][s[i]] | T.dec[1][s[i + 1]] | T.dec[2][s[i + 2]] | T.dec[3][s[i + 3]];
        if (v & 0x80000000u) return -1;
        o[0] = (uint8_t)(v >> 16);
        o[1] = (uint8_t)(v >> 8);
        o[2] = (uint8_t)v;
    }
    const int64_t rem = n - i;
    if (rem == 1) return -1;
    if (rem >= 2) {
        const uint32_t v = T.dec[0][s[i]] | T.dec[1][s[i + 1]] | (rem == 3 ? T.dec[2][s[i + 2]] : 0u);
        if (v & 0x80000000u) return -1;
        *o++ = (uint8_t)(v >> 16);
        if (rem == 3) *o++ = (uint8_t)(v >> 8);
    }
    return (int64_t)(o - out);
}

}  // extern "C"
