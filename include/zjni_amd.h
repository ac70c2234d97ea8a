/* zjni_amd.h — C-ABI of the MI355X-native zstd path for zstd-jni (libzjni_amd.so).
 *
 * Boundary: these entry points are what the reference's JNI glue
 * (/root/reference/src/main/native/jni_fast_zstd.c, jni_zstd.c; "N/" below) would bind instead of
 * calling libzstd's ZSTD_compress2 / ZSTD_decompressDCtx for batches of independent buffers.
 * Plain pointers and sizes only; no torch / HIP types in any signature (streams are passed as void*).
 *
 * Result convention everywhere = the reference's: a size_t byte count, or (size_t)(0 - code) with
 * `code` a ZSTD_ErrorCode (N/zstd_errors.h:60-98); test with zjni_isError (== N/jni_zstd.c:240-243
 * Zstd.isError) and map with zjni_getErrorCode / zjni_getErrorName (== Zstd.getErrorCode /
 * getErrorName, N/jni_zstd.c:252-267).
 *
 * The library REQUIRES a gfx950 device: every compute entry returns ZJNI_ERROR(no_device) (code 200,
 * outside libzstd's range) when HIP finds none.  There is no CPU fallback inside this library; the
 * JNI glue keeps libzstd for what the GPU path does not cover.
 */
#ifndef ZJNI_AMD_H
#define ZJNI_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZJNI_VERSION_STRING "0.1.0-zstd1.5.7"
#define ZJNI_ERROR_no_device 200u      /* no HIP device / kernel launch failure */
#define ZJNI_ERROR_unsupported 201u    /* input outside the GPU path's scope (see zjni_compress_batch) */
#define ZJNI_BLOCKSIZE_MAX (1u << 17)  /* ZSTD_BLOCKSIZE_MAX, N/zstd.h:147-148: inputs up to here become single-block frames */
#define ZJNI_FRAME_MAX (2u << 20)      /* largest input the compress entries take: multi-block frames (N/compress/zstd_compress.c:4591-4692),
                                        * byte-identical to ZSTD_compress2 with the level's own parameters, as long as the frame fits the level's
                                        * window — 512 KiB / 1 MiB / 2 MiB at levels 1 / 2 / 3; beyond that ZJNI_ERROR_unsupported */
#define ZJNI_LEVEL4_MAX (1u << 17)     /* level 4 (N/compress/clevels.h:84,110): greedy on the hash chain up to 16 KiB (ZSTD_compressBlock_greedy,
                                        * N/compress/zstd_lazy.c:1784), double-fast with 2^17-entry tables up to here; larger inputs run the
                                        * reference's row-based match finder, which this library does not restate: ZJNI_ERROR_unsupported */
#define ZJNI_LAZY_MAX (1u << 17)       /* levels 5-8 (N/compress/clevels.h:85-88,111-114): greedy / lazy / lazy2 (ZSTD_compressBlock_lazy_generic,
                                        * N/compress/zstd_lazy.c:1516-1780) on the hash chain for inputs up to 16 KiB and on the row-based finder
                                        * (ZSTD_RowFindBestMatch, :1141) above, one block per frame */

/* ---- library / device ---- */
const char* zjni_version(void);
/* Number of usable HIP devices (0 when none). */
int zjni_device_count(void);
/* Selects device `ordinal` for the calling thread's subsequent calls and creates its per-device
 * state (work counters, literal scratch). Returns 0 or a negative ZJNI/ZSTD error code. */
int zjni_init(int ordinal);
void zjni_shutdown(void);

/* ---- error helpers: N/jni_zstd.c:240-267 (Zstd.isError / getErrorName / getErrorCode) ---- */
unsigned zjni_isError(size_t result);
int zjni_getErrorCode(size_t result);
const char* zjni_getErrorName(size_t result);

/* ---- sizing helpers ---- */
/* == ZSTD_compressBound (N/zstd.h:249) == Zstd.compressBound (N/jni_zstd.c:229-232) */
size_t zjni_compressBound(size_t srcSize);
/* == ZSTD_getFrameContentSize (N/zstd.h:203-217) == Zstd.getFrameContentSize0 (N/jni_zstd.c:116-130):
 * content size, (uint64)-1 unknown, (uint64)-2 error.  Host-side header parse, no GPU needed. */
unsigned long long zjni_getFrameContentSize(const void* src, size_t srcSize);

/* ---- hot path, device-resident batch -------------------------------------------------------
 * HBM layout: n independent buffers packed in one blob; buffer i occupies
 * [d_src + d_src_off[i], d_src + d_src_off[i+1]) and may write
 * [d_dst + d_dst_off[i], d_dst + d_dst_off[i+1]).  Offsets are uint64[n+1] in device memory.
 * d_result[i] receives buffer i's produced size or error (uint64, same convention as size_t).
 * Asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream).
 * Returns 0 when the launch was enqueued, else an error code result. */

/* Replaces ZSTD_decompressDCtx (N/jni_fast_zstd.c:798-799, :825-826) for n frames at once.
 * Every source buffer may hold several concatenated frames and skippable frames, exactly as
 * ZSTD_decompressMultiFrame accepts (N/decompress/zstd_decompress.c:1070-1168). */
size_t zjni_decompress_batch_device(const void* d_src, const uint64_t* d_src_off,
                                    void* d_dst, const uint64_t* d_dst_off,
                                    uint64_t* d_result, size_t n, void* stream);

/* Replaces ZSTD_CCtx_reset + ZSTD_compress2 (N/jni_fast_zstd.c:606-607, :633-635) for n buffers at
 * once: each buffer becomes one standard zstd frame (content size in the header, no checksum,
 * no dictID) that any zstd decoder accepts.  level: 1..3 (N/compress/clevels.h), or 4 for inputs up to ZJNI_LEVEL4_MAX / 5..8 up to ZJNI_LAZY_MAX (plain
 * entries only: no dictionary, no explicit table sizes; a wave-per-frame kernel with one lane parsing — exact, not fast).  Buffers larger than
 * ZJNI_BLOCKSIZE_MAX become multi-block frames (one wavefront per frame, block after block; see ZJNI_FRAME_MAX for the
 * range); beyond it d_result[i] reports ZJNI_ERROR_unsupported and the buffer stays on the CPU path.
 * Destination capacity (d_dst_off[i+1] - d_dst_off[i]): with zjni_compressBound(srcSize) a frame always fits.  With less, the answer is
 * ZSTD_compress2's for that capacity — which wants working room beyond the frame's bytes (18 bytes for any header, 8 bytes of slack behind
 * each bit stream, N/compress/zstd_compress.c:4711, 3024-3030; DESIGN.md section 1 "tight destination"): ZSTD_error_dstSize_tooSmall possibly
 * although the frame would have fitted, or a raw block where the compressed one found no room — same size, bytes or code. */
size_t zjni_compress_batch_device(const void* d_src, const uint64_t* d_src_off,
                                  void* d_dst, const uint64_t* d_dst_off,
                                  uint64_t* d_result, size_t n, int level, void* stream);
/* Same with ZSTD_c_checksumFlag = checksum (ZstdCompressCtx.setChecksum0, N/jni_fast_zstd.c:276-282;
 * Zstd.compressUnsafe(..., checksumFlag), N/jni_zstd.c:50-63): frames end with the low 32 bits of
 * XXH64(content, 0).  The decompress entries verify a frame's checksum whenever its header announces one
 * (ZSTD_error_checksum_wrong on mismatch), like ZSTD_decompressDCtx. */
size_t zjni_compress_batch_device2(const void* d_src, const uint64_t* d_src_off,
                                   void* d_dst, const uint64_t* d_dst_off,
                                   uint64_t* d_result, size_t n, int level, int checksum, void* stream);

/* ---- dictionaries, decompress side (SURVEY.md §8a last rows; BASELINE config 4) ----
 * zjni_ddict == ZSTD_DDict as zstd-jni holds it in ZstdDictDecompress.nativePtr (J/ZstdDictDecompress.java,
 * N/jni_fast_zstd.c:56-96): created once from the dictionary bytes (zstd dictionary format with magic
 * 0xEC30A437, or raw content), shared read-only by any number of batch calls on the device it was created on.
 * createDDict returns NULL for a corrupted dictionary (ZSTD_createDDict does the same) or without a device. */
typedef struct zjni_ddict zjni_ddict;
zjni_ddict* zjni_createDDict(const void* dict, size_t dictSize);
size_t zjni_freeDDict(zjni_ddict* ddict);
unsigned zjni_getDictID_fromDDict(const zjni_ddict* ddict);
/* Replaces ZSTD_decompress_usingDDict (N/jni_fast_zstd.c:133-183 decompress*FastDict0, and
 * ZstdDecompressCtx.loadDict + decompress*0) for n frames at once.  ddict == NULL behaves like
 * zjni_decompress_batch_device.  Frames naming another dictionary ID report ZSTD_error_dictionary_wrong. */
size_t zjni_decompress_batch_device_usingDDict(const void* d_src, const uint64_t* d_src_off,
                                               void* d_dst, const uint64_t* d_dst_off,
                                               uint64_t* d_result, size_t n, const zjni_ddict* ddict, void* stream);
size_t zjni_decompress_batch_usingDDict(const void* const* src, const size_t* srcSize,
                                        void* const* dst, const size_t* dstCapacity,
                                        size_t* result, size_t n, const zjni_ddict* ddict);
size_t zjni_decompress_usingDDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zjni_ddict* ddict);

/* ---- explicit table sizes: ZstdCompressCtx.setHashLog / setChainLog (J/ZstdCompressCtx.java; N/jni_fast_zstd.c setHashLog0 /
 * setChainLog0 -> ZSTD_c_hashLog / ZSTD_c_chainLog) on top of level + checksum; 0 = not set.  Honoured for level 3
 * (double-fast): hashLog 6..17, chainLog 6..16, frames byte-identical to the reference called with the same two
 * parameters.  Not set, level 3 uses the reference's own sizes for the input (16 / 15 at 64 KiB: the frames of a plain
 * Zstd.compress(x, 3)); 14 / 13 are the sizes the LDS-resident finders of small batches are built for, honoured when asked for.
 * Other levels with a non-zero value: ZSTD_error_parameter_unsupported; out of range: parameter_outOfBound. */
/* Frame-header parameters: the `checksum` argument of the *_advanced and *_usingCDict entries is a flag word —
 * ZSTD_c_checksumFlag, ZSTD_c_contentSizeFlag = 0 (ZstdCompressCtx.setContentSize0(false), N/jni_fast_zstd.c:301-308: no frame
 * content size, a window descriptor instead; without a dictionary only) and ZSTD_c_dictIDFlag = 0 (setDictID0(false), :313-320).
 * A plain 0 / 1 keeps meaning "checksum off / on" with the other two at their defaults. */
#define ZJNI_FRAME_CHECKSUM       1
#define ZJNI_FRAME_NO_CONTENTSIZE 2
#define ZJNI_FRAME_NO_DICTID      4
size_t zjni_compress_batch_device_advanced(const void* d_src, const uint64_t* d_src_off,
                                           void* d_dst, const uint64_t* d_dst_off,
                                           uint64_t* d_result, size_t n, int level, int checksum, int hashLog, int chainLog, void* stream);
size_t zjni_compress_batch_advanced(const void* const* src, const size_t* srcSize,
                                    void* const* dst, const size_t* dstCapacity,
                                    size_t* result, size_t n, int level, int checksum, int hashLog, int chainLog);

/* ---- stream frames: ZstdDirectBufferCompressingStream[NoFinalizer] / ZstdOutputStream[NoFinalizer] ----
 * Replaces ZSTD_compressStream / ZSTD_flushStream / ZSTD_endStream as the stream natives call them
 * (N/jni_directbuffercompress_zstd.c:97-161: compressDirectByteBuffer, flushStream, endStream; N/jni_outputstream_zstd.c) for a stream that is
 * BUFFERED UNTIL IT IS CLOSED, at most the level's unknown-size window (levels 1 / 2 / 3: 512 KiB / 1 MiB / 2 MiB): the frame ZSTD_compressStream2
 * produces without a pledged size, byte for byte (N/compress/zstd_compress.c:6103-6300, :4591-4692) — the level's default parameter row whatever
 * the total, no content size in the header, the input cut into the stream's 128 KiB pieces, blocks ended where the caller flushed, the empty raw
 * last block, the checksum.  flushAt[0 .. nFlush): ascending byte counts after which the caller flushed.  final_ = 0: flushed, not closed — returns the
 * frame's beginning up to the last flush (the bytes a later call with more input reproduces and continues).  knownEmpty: the stream was closed before
 * any other call (its size, 0, is then known: single-segment header).  201 above the window, 42 above level 3: the bundled library's stream.
 * The device form takes n streams: stream i = blob[off[i], off[i + 1]), its flush positions d_flush_at[d_flush_off[i] .. d_flush_off[i + 1]) (d_flush_off may be
 * NULL: none), d_mode[i] = final | knownEmpty << 1 (NULL: all final). */
/* ZSTD_findFrameCompressedSize + ZSTD_getFrameContentSize + ZSTD_decompressBound for one complete zstd frame at src (N/zstd.h:227-290;
 * N/decompress/zstd_decompress.c:739-850) — what the decompress-stream natives (N/jni_directbufferdecompress_zstd.c:58-79) need to know before they can hand a
 * whole frame to zjni_decompress instead of feeding ZSTD_decompressStream: returns the frame's size in bytes (0: src does not hold a complete well-formed zstd
 * frame), *content = the recorded content size or ~0, *bound = an upper bound of the decoded size. */
size_t zjni_frame_extent(const void* src, size_t srcSize, unsigned long long* content, unsigned long long* bound);
size_t zjni_compress_stream(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksum,
                            const uint32_t* flushAt, size_t nFlush, int final_, int knownEmpty);
size_t zjni_compress_stream_batch_device(const void* d_src, const uint64_t* d_src_off, void* d_dst, const uint64_t* d_dst_off,
                                         uint64_t* d_result, size_t n, int level, int checksum,
                                         const uint32_t* d_flush_at, const uint64_t* d_flush_off, const uint32_t* d_mode, void* stream);

/* ---- compression dictionaries ----
 * zjni_cdict == ZSTD_CDict as zstd-jni holds it in ZstdDictCompress.nativePtr (J/ZstdDictCompress.java;
 * N/jni_fast_zstd.c:18-66: init = ZSTD_createCDict(dict, size, level), free = ZSTD_freeCDict).  The dictionary is
 * digested once on the device: parameters of (level, dictSize), tagged hash tables over the content, the entropy
 * tables and repcodes of its header.  Levels 1..3.  NULL for a corrupted dictionary, a bad level, fewer than
 * 8 bytes, or without a device. */
typedef struct zjni_cdict zjni_cdict;
zjni_cdict* zjni_createCDict(const void* dict, size_t dictSize, int level);
size_t zjni_freeCDict(zjni_cdict* cdict);
unsigned zjni_getDictID_fromCDict(const zjni_cdict* cdict);
/* Replaces ZstdCompressCtx.loadDict(ZstdDictCompress) + compress*0 = ZSTD_CCtx_refCDict + ZSTD_compress2
 * (N/jni_fast_zstd.c:325-336, :586-640) and ZSTD_compress_usingCDict (compress*FastDict0, N/jni_fast_zstd.c:171-216)
 * for n buffers at once; frames are byte-identical to the reference's.  Covers the sizes at which the reference
 * searches the dictionary in place ("attach": srcSize <= 8 KiB when the dictionary's strategy is fast, <= 16 KiB when
 * it is double-fast) and, beyond them up to one block (128 KiB), its copy mode (ZSTD_resetCCtx_byCopyingCDict,
 * N/compress/zstd_compress.c:2402-2468: the dictionary as an external segment) as long as the reference compresses with
 * the dictionary's own parameters (srcSize < 128 KiB or < 6 x the dictionary's content, :5256-5257); otherwise the
 * result slot reports ZSTD_error_parameter_unsupported (more than one block: ZJNI_ERROR_unsupported). */
size_t zjni_compress_batch_device_usingCDict(const void* d_src, const uint64_t* d_src_off,
                                             void* d_dst, const uint64_t* d_dst_off,
                                             uint64_t* d_result, size_t n, const zjni_cdict* cdict, int checksum, void* stream);
size_t zjni_compress_batch_usingCDict(const void* const* src, const size_t* srcSize,
                                      void* const* dst, const size_t* dstCapacity,
                                      size_t* result, size_t n, const zjni_cdict* cdict, int checksum);
size_t zjni_compress_usingCDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zjni_cdict* cdict);

/* ---- hot path, host buffers (what a JNI batch native binds; stages through pinned memory) ---- */
size_t zjni_decompress_batch(const void* const* src, const size_t* srcSize,
                             void* const* dst, const size_t* dstCapacity,
                             size_t* result, size_t n);
size_t zjni_compress_batch(const void* const* src, const size_t* srcSize,
                           void* const* dst, const size_t* dstCapacity,
                           size_t* result, size_t n, int level);
size_t zjni_compress_batch2(const void* const* src, const size_t* srcSize,
                            void* const* dst, const size_t* dstCapacity,
                            size_t* result, size_t n, int level, int checksum);

/* ---- the same entries, asynchronous: two host batches in flight (round 5) ----
 * zstd-jni's natives block (N/jni_fast_zstd.c:586-640: one ZSTD_compress2 per call); a batch native over this library would too, and a single host batch is
 * a chain: gather + H2D, kernels, D2H + scatter (75 + 143 + 35 ms on 65 536 x 64 KiB at level 3) — the lane pipeline wants the whole batch resident before its
 * kernels start.  The overlap comes from the NEXT batch: every device has two staging slots with their own streams, so while one call's kernels run the other
 * call's sources cross the link one way and a finished call's frames the other.  Two threads inside zjni_compress_batch2 / zjni_decompress_batch get that by
 * themselves; these entries give it to one thread: _begin returns at once with a job that runs the blocking entry on a thread of the library (bound to the
 * caller's device), zjni_batch_finish waits for it, returns the call's code and frees the job.  Every array and buffer passed to _begin belongs to the job until
 * _finish returns; `result` is written as in the blocking entries.  NULL: no device bound, or no thread to be had.  More than two jobs may be begun; the third
 * waits for a slot.  What compressBatch0 / decompressBatch0 of the JNI library would call to keep two Java-side batches in flight. */
typedef struct zjni_batch_job zjni_batch_job;
zjni_batch_job* zjni_compress_batch_begin(const void* const* src, const size_t* srcSize,
                                          void* const* dst, const size_t* dstCapacity,
                                          size_t* result, size_t n, int level, int checksum);
zjni_batch_job* zjni_decompress_batch_begin(const void* const* src, const size_t* srcSize,
                                            void* const* dst, const size_t* dstCapacity,
                                            size_t* result, size_t n);
size_t zjni_batch_finish(zjni_batch_job* job);

/* ---- one host batch over several GPUs of this process (a JVM is one process) ----
 * The batch is cut into contiguous index ranges of about equal source bytes, one per entry of `devices` (ordinals, nDevices <= 64;
 * an ordinal may repeat); one thread per device runs its range.  Frames and results are what the single-device entries give.
 * mode 0: every device returns its frames to the host over its own PCIe link — no inter-GPU traffic, the right choice for host
 *         consumers (JVM buffers).
 * mode 1: (compress) every device packs its frames and sends them to devices[0] over xGMI (peer copies: the output gather of
 *         section 8e, 7 links into one device), which returns the whole batch in one transfer.
 * The calling thread's own device binding is left as it was. */
size_t zjni_compress_batch_multi(const void* const* src, const size_t* srcSize,
                                 void* const* dst, const size_t* dstCapacity,
                                 size_t* result, size_t n, int level, int checksum,
                                 const int* devices, int nDevices, int mode);
size_t zjni_decompress_batch_multi(const void* const* src, const size_t* srcSize,
                                   void* const* dst, const size_t* dstCapacity,
                                   size_t* result, size_t n, const int* devices, int nDevices);

/* ---- cross-thread aggregation of per-buffer calls ----
 * zstd-jni's per-buffer natives (N/jni_fast_zstd.c:586-640, :777-905) are called from many threads, one buffer each.  An aggregator
 * turns concurrent blocking calls into batches: the first caller of a kind (compress at a level + checksum flag / decompress) opens a
 * batch and waits up to maxWaitMicros (or until maxBatch callers have joined), then runs zjni_compress_batch2 / zjni_decompress_batch
 * once for everybody.  Each call returns what the per-buffer form would (frame size or error code).  stats: calls made, batches run. */
typedef struct zjni_aggregator zjni_aggregator;
zjni_aggregator* zjni_createAggregator(int device, size_t maxBatch, unsigned maxWaitMicros);
void zjni_freeAggregator(zjni_aggregator* a);
size_t zjni_aggregator_compress(zjni_aggregator* a, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksum);
size_t zjni_aggregator_decompress(zjni_aggregator* a, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
void zjni_aggregator_stats(zjni_aggregator* a, unsigned long long* calls, unsigned long long* batches);

/* ---- per-buffer forms with the exact argument meaning of the calls they replace ---- */
/* ZSTD_compress2(cctx{level}, dst, dstCapacity, src, srcSize): N/jni_fast_zstd.c:607 */
size_t zjni_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level);
/* the same with ZSTD_c_checksumFlag (Zstd.compress(dst, src, level, checksumFlag), J/Zstd.java) */
size_t zjni_compress2(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksum);
/* ZSTD_decompressDCtx(dctx, dst, dstCapacity, src, srcSize): N/jni_fast_zstd.c:799 */
size_t zjni_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);

/* ---- synthetic mixed-entropy workload, identical bytes on host and device ---- */
void zjni_synth_fill_host(void* dst, size_t bufSize, uint64_t firstIndex, size_t nBuffers);
size_t zjni_synth_fill_device(void* d_dst, size_t bufSize, uint64_t firstIndex, size_t nBuffers, void* stream);

/* ---- output assembly for the multi-GPU gather ----
 * Moves frame i from d_src[d_src_off[i] .. +d_sizes[i]) to d_dst[d_dst_off[i] ..): tight packing of a
 * compress batch's variable-size outputs before the RCCL payload gather.  Entries of d_sizes that are
 * error results are skipped. */
size_t zjni_pack_batch_device(const void* d_src, const uint64_t* d_src_off, const uint64_t* d_sizes,
                              void* d_dst, const uint64_t* d_dst_off, size_t n, void* stream);
/* The same with the packed offsets made on the device: d_packed_off[0 .. n] (out) = exclusive prefix sums of the sizes (error results count as 0),
 * then the frames move there.  One call instead of a scan by the caller plus the pack (round 5). */
size_t zjni_pack_batch_device2(const void* d_src, const uint64_t* d_src_off, const uint64_t* d_sizes,
                               void* d_dst, uint64_t* d_packed_off, size_t n, void* stream);

/* ---- resource policy ----
 * The large-batch pipelines keep per-frame scratch in HBM (match-finder tables and sequence records, decode cells): one buffer
 * per pipeline and device, allocated on first use and kept, sized for a 288 GB part (65 536 frames in flight: ~26 GiB compress,
 * ~49 GiB for frames > 64 KiB, ~34 GiB with a dictionary, ~22 GiB decompress).  A process that shares the GPU bounds that:
 *   zjni_set_scratch_limit(bytes)  total scratch the library may hold per device (0 = no limit; values below 4 GiB are raised to
 *                                  4 GiB).  Batches are cut into slices whose scratch fits — results are unchanged, throughput
 *                                  drops as slices shrink — and a pipeline that needs room evicts the others' buffers.
 *                                  Returns the limit in force.  The limit is process-wide.
 *   zjni_scratch_bytes()           scratch currently held on the calling thread's device.
 *   zjni_release_scratch()         free-on-idle: waits for the device's outstanding batch calls, then frees all of it (the next
 *                                  call allocates again).  0 or an error code.
 * Concurrent callers of one device are serialised: batch calls are enqueued under a per-device mutex and each call's first kernel
 * waits (on the GPU) for the previous call's last one, whatever streams the callers use; the scratch is shared, results are not. */
size_t zjni_set_scratch_limit(size_t bytes);
size_t zjni_scratch_bytes(void);
size_t zjni_release_scratch(void);

/* ---- introspection for tests/bench ---- */
/* Workgroups the persistent kernels launch per device and LDS bytes per workgroup. */
int zjni_kernel_info(int* decodeGrid, int* decodeLdsBytes, int* encodeGrid, int* encodeLdsBytes);
/* Stage durations (ms) of the last large-batch device calls, from HIP events recorded on the caller's stream
 * around the kernels: out5[0] lane-per-frame match-finder kernel (compress); out5[1..4] decode stages prep /
 * lane-per-frame sequence decode / execute / fused leftovers.  Blocks until those stages have completed;
 * stages that did not run read -1.  Profiling aid (bench.py roofline), not on the data path. */
int zjni_last_timing(float* out5);
/* The same five plus out8[5] = the wide match-finder kernel (frames > 64 KiB), last slice of the call; out8[6..7] read -1. */
int zjni_last_timing2(float* out8);
/* Which match finder served list A of the last large compress call on this device (bench.py names the roofline's kernel from this, not from the
 * environment: a refused LDS attribute, a scratch budget or a failed allocation changes the route silently).  Negative: no device. */
#define ZJNI_ROUTE_NONE 0         /* no large-batch compress call yet */
#define ZJNI_ROUTE_FUSED 1        /* zj_encode_kernel alone (small batches, levels 1-2) */
#define ZJNI_ROUTE_WAVE 2         /* zj_enc_match_wave_kernel (small level-3 batches, tables in LDS) */
#define ZJNI_ROUTE_LANE 3         /* zj_enc_match_kernel (levels 1-2; level 3 under ZJNI_LANE_MACHINE=0 without flags) */
#define ZJNI_ROUTE_LANE_GATED 4   /* zj_enc_match_gated_kernel (ZJNI_LANE_MACHINE=0 with flags) */
#define ZJNI_ROUTE_RUN 5          /* zj_enc_match_run_kernel without need flags */
#define ZJNI_ROUTE_RUN_FLAGS 6    /* zj_enc_match_run_kernel behind zj_enc_worth_kernel / zj_enc_need_kernel */
#define ZJNI_ROUTE_HYBRID 7       /* ZJNI_HYBRID=1: lane and wave kernels side by side */
#define ZJNI_ROUTE_OTHER 8        /* levels 4-8, dictionaries, multi-block only */
#define ZJNI_ROUTE_WAVE_HBM 9     /* zj_encode_multi_kernel: level-3 frames of batches below ZJNI_L3_WAVE_MAX, wave per frame over HBM tables (zj_match_wavex.h) */
#define ZJNI_ROUTE_WIDE 10        /* zj_enc_match_wide_kernel: the launch of frames above 64 KiB (list B); never zjni_last_route()'s answer — see zjni_last_lists */
#define ZJNI_ROUTE_PIPE 11        /* zj_encode_pipe_kernel (round 6): multi-block frames of a batch that leaves wave slots empty, a parse wave a block ahead of an entropy wave per frame;
                                     decided on the device from the list counts, so zjni_last_route() says it only after zjni_last_lists() has read them */
int zjni_last_route(void);
/* How the last large compress call's frames were split: out3[0] the common launch (the route above), out3[1] the wide launch (ZJNI_ROUTE_WIDE), out3[2] the
 * multi-block / wave-per-frame kernel.  A batch of 128 KiB buffers has out3[1] = n: its match-finder time is zjni_last_timing2's out8[5] and its kernel
 * zjni_route_kernel(ZJNI_ROUTE_WIDE).  Synchronises with the device (diagnostics only). */
int zjni_last_lists(unsigned* out3);
/* The same for the last large decompress call: out4[0] frames of the single-block pipeline (prep -> lane-per-frame sequence decode -> execute), out4[1] frames the
 * fused wave-per-frame kernel decoded (what no pipeline took, or handed over), out4[2] frames of the multi-block stages (lane per BLOCK; frames of several blocks
 * or without a content size), out4[3] their blocks.  Synchronises with the device (diagnostics only). */
int zjni_last_decode_lists(unsigned* out4);
/* zjni_last_decode_lists plus out5[4]: frames that are a header and ONE stored (raw or RLE) block — what ZSTD_compress2 writes for data that does not compress
 * (N/compress/zstd_compress.c:4591-4692, ZSTD_noCompressBlock) — which stage 1 of the large-batch pipeline copies itself; they are on none of the three lists. */
int zjni_last_decode_lists2(unsigned* out5);
/* Name of the kernel a route's match-finder time (zjni_last_timing out[0]) belongs to. */
const char* zjni_route_kernel(int route);
/* The source revision the library was built from ("unknown" when the build had no git): profiles and PMC passes are stamped with it. */
const char* zjni_build_stamp(void);

#ifdef __cplusplus
}
#endif
#endif /* ZJNI_AMD_H */
