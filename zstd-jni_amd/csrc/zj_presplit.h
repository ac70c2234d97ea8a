// zj_presplit.h — where a full 128 KiB block of a multi-block frame is cut: ZSTD_splitBlock (N/compress/zstd_preSplit.c:152-238),
// as ZSTD_optimalBlockSize calls it (N/compress/zstd_compress.c:4552-4581) for the strategies of levels 1-3:
//   fast   -> split level 0: ZSTD_splitBlock_fromBorders — byte histograms of the first, the last and the middle 512 bytes
//   dfast  -> split level 1: ZSTD_splitBlock_byChunks(level 0) — byte histograms of every 43rd byte of 8 KiB chunks, cut where a
//             chunk differs from everything before it
// Serial (one lane); `ev` is scratch for three 256-entry histograms.  Returns the size of the next block.
#pragma once

ZJ_DEV u64 zp_abs64(i64 v) { return (u64)(v < 0 ? -v : v); }
// fpDistance over 256 events (hashLog 8): sum |a[n] * nb - b[n] * na|
ZJ_DEV u64 zp_distance(const u32* a, u32 na, const u32* b, u32 nb) {
    u64 d = 0;
    for (u32 n = 0; n < 256; n++) d += zp_abs64((i64)a[n] * (i64)nb - (i64)b[n] * (i64)na);
    return d;
}
// compareFingerprints: 1 when "too different"
ZJ_DEV bool zp_differ(const u32* ref, u32 nref, const u32* nw, u32 nnew, u32 penalty) {
    u64 const p50 = (u64)nref * (u64)nnew;
    u64 const threshold = p50 * (u64)(14u + penalty) / 16u;            // THRESHOLD_BASE = 16 - 2, THRESHOLD_PENALTY_RATE = 16
    return zp_distance(ref, nref, nw, nnew) >= threshold;
}
ZJ_DEV u32 zp_split_from_borders(const u8* p, u32* ev) {               // blockSize == 128 KiB
    u32* const first = ev; u32* const last = ev + 256; u32* const mid = ev + 512;
    for (u32 n = 0; n < 768; n++) ev[n] = 0;
    for (u32 n = 0; n < 512; n++) { first[p[n]]++; last[p[131072u - 512u + n]]++; }
    if (!zp_differ(first, 512, last, 512, 0)) return 131072u;
    for (u32 n = 0; n < 512; n++) mid[p[65536u - 256u + n]]++;
    u64 const fromBegin = zp_distance(first, 512, mid, 512), fromEnd = zp_distance(last, 512, mid, 512);
    u64 const minDistance = 512u * 512u / 3u;
    if (zp_abs64((i64)fromBegin - (i64)fromEnd) < minDistance) return 65536u;
    return fromBegin > fromEnd ? 32768u : 98304u;
}
ZJ_DEV u32 zp_split_by_chunks(const u8* p, u32* ev) {                  // level 0: sampling rate 43, hashLog 8 (the byte itself)
    u32* const past = ev; u32* const nw = ev + 256;
    u32 const perChunk = (8192u - 2u + 1u) / 43u;                      // limit / samplingRate events per 8 KiB chunk
    u32 nPast, penalty = 3;
    for (u32 n = 0; n < 256; n++) past[n] = 0;
    for (u32 n = 0; n < 8192u - 1u; n += 43u) past[p[n]]++;
    nPast = perChunk;
    for (u32 pos = 8192u; pos <= 131072u - 8192u; pos += 8192u) {
        for (u32 n = 0; n < 256; n++) nw[n] = 0;
        for (u32 n = 0; n < 8192u - 1u; n += 43u) nw[p[pos + n]]++;
        if (zp_differ(past, nPast, nw, perChunk, penalty)) return pos;
        for (u32 n = 0; n < 256; n++) past[n] += nw[n];
        nPast += perChunk;
        if (penalty > 0) penalty--;
    }
    return 131072u;
}
// zp_split_by_chunks on the whole group: the 191 samples of a chunk are loaded by the lanes at once (one lane asks for them one round
// trip after the other: 1.8 M cycles per block, measured), histograms by LDS atomics, the distance summed over the lanes — integer
// arithmetic throughout, the same value whatever the order.  `ev`: 256 + 256 histogram words and 2 words for the sum, in LDS.
template <class G>
ZJ_DEV u32 zp_split_by_chunks_g(const G& g, const u8* p, u32* ev) {
    u32* const past = ev; u32* const nw = ev + 256; u32* const sum = ev + 512;      // sum: low / high word of the 64-bit distance
    u32 const perChunk = (8192u - 2u + 1u) / 43u, samples = (8192u - 1u + 42u) / 43u;
    u32 nPast = perChunk, penalty = 3;
    GRP_FOR(g, n, 256) past[n] = 0;
    g.sync();
    GRP_FOR(g, i, samples) atomicAdd(&past[p[i * 43u]], 1u);
    g.sync();
    for (u32 pos = 8192u; pos <= 131072u - 8192u; pos += 8192u) {
        GRP_FOR(g, n, 256) nw[n] = 0;
        GRP_SERIAL(g) { sum[0] = 0; sum[1] = 0; }
        g.sync();
        GRP_FOR(g, i, samples) atomicAdd(&nw[p[pos + i * 43u]], 1u);
        g.sync();
        {   u64 d = 0;
            GRP_FOR(g, n, 256) d += zp_abs64((i64)past[n] * (i64)perChunk - (i64)nw[n] * (i64)nPast);
            // (d < 2^8 entries x 2^17 x 2^12: the low words' carries are counted into the high word)
            u32 const lo = (u32)d, hi = (u32)(d >> 32);
            u32 const before = atomicAdd(&sum[0], lo);
            atomicAdd(&sum[1], hi + ((u32)(before + lo) < lo ? 1u : 0u)); }
        g.sync();
        u64 const dist = ((u64)ZJ_UNI(sum[1]) << 32) | ZJ_UNI(sum[0]);
        u64 const threshold = (u64)nPast * (u64)perChunk * (u64)(14u + penalty) / 16u;
        g.sync();
        if (dist >= threshold) return pos;
        GRP_FOR(g, n, 256) past[n] += nw[n];
        nPast += perChunk;
        if (penalty > 0) penalty--;
        g.sync();
    }
    return 131072u;
}
// ZSTD_optimalBlockSize for levels 1-3 (blockSizeMax = 128 KiB, default pre-split level)
ZJ_DEV u32 zp_block_size(const u8* p, u32 remaining, u32 strategy, i64 savings, u32* ev) {
    if (remaining < 131072u) return remaining;
    if (savings < 3) return 131072u;
    return strategy == 1 ? zp_split_from_borders(p, ev) : zp_split_by_chunks(p, ev);
}
