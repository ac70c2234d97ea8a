// zj_synth.h — deterministic "mixed-entropy" buffer generator of SURVEY.md §8(d) / BASELINE.md §3.
// One definition shared by the host (tests, CPU baseline) and the device (bench data is generated
// in HBM, never copied from the host).
//   PRNG   xorshift64, per-buffer seed = 0x9E3779B97F4A7C15 ^ ((index+1) * 0xD6E8FEB86659FD93), 4 warm-up steps
//   class  index & 3 : 0 text (16-word vocabulary), 1 JSON-like records, 2 low-entropy binary
//          (7/8: copy a byte from distance 1..64, else a random 4-bit value), 3 uniform random bytes
#pragma once
#include "zj_common.h"

ZJ_HD u64 zs_next(u64& x) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }

ZJ_HD u32 zs_word(u32 w, char* out) {   // copies vocabulary word w (0..15), returns its length
    const char* const voc[16] = { "the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog",
                                  "compression", "wavefront", "dictionary", "sequence", "literal", "offset", "window", "entropy" };
    const char* s = voc[w & 15]; u32 n = 0;
    while (s[n]) { out[n] = s[n]; n++; }
    return n;
}
ZJ_HD u32 zs_dec(u32 v, char* out) {    // decimal, returns length
    char tmp[10]; u32 n = 0, k;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (k = 0; k < n; k++) out[k] = tmp[n - 1 - k];
    return n;
}

ZJ_HD void zs_fill(u8* out, u32 n, u64 index) {
    u64 x = 0x9E3779B97F4A7C15ull ^ ((index + 1) * 0xD6E8FEB86659FD93ull);
    if (x == 0) x = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; i++) zs_next(x);
    u32 const cls = (u32)(index & 3);
    u32 pos = 0;
    if (cls == 0) {
        char w[16];
        while (pos < n) {
            u32 const len = zs_word((u32)(zs_next(x) >> 20), w);
            for (u32 k = 0; k < len && pos < n; k++) out[pos++] = (u8)w[k];
            if (pos < n) out[pos++] = ' ';
        }
    } else if (cls == 1) {
        char rec[64]; u32 id = (u32)(index * 1000u);
        while (pos < n) {
            u64 const r = zs_next(x); u32 m = 0;
            const char a[] = "{\"id\":"; for (u32 k = 0; k < 6; k++) rec[m++] = a[k];
            m += zs_dec(id++, rec + m);
            const char b[] = ",\"name\":\""; for (u32 k = 0; k < 9; k++) rec[m++] = b[k];
            m += zs_word((u32)(r >> 8), rec + m);
            const char c[] = "\",\"v\":"; for (u32 k = 0; k < 6; k++) rec[m++] = c[k];
            m += zs_dec((u32)(r >> 32) % 100000u, rec + m);
            rec[m++] = '}'; rec[m++] = ',';
            for (u32 k = 0; k < m && pos < n; k++) out[pos++] = (u8)rec[k];
        }
    } else if (cls == 2) {
        while (pos < n) {
            u64 const r = zs_next(x);
            if ((r & 7) != 0 && pos > 0) {
                u32 d = 1 + (u32)((r >> 3) & 63);
                if (d > pos) d = pos;
                out[pos] = out[pos - d];
            } else out[pos] = (u8)((r >> 9) & 15);
            pos++;
        }
    } else {
        while (pos < n) {
            u64 r = zs_next(x);
            for (u32 k = 0; k < 8 && pos < n; k++) { out[pos++] = (u8)r; r >>= 8; }
        }
    }
}
