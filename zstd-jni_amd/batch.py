"""Device-resident batches: the HBM layout the kernels consume (include/zjni_amd.h).

A batch is one uint8 blob in HBM plus an int64[n+1] offsets tensor (buffer i = blob[off[i]:off[i+1]]).
torch is used only for device memory and the stream handle; all compute is in libzjni_amd.so.
"""
import torch

from . import lib, ZstdException


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _check(r):
    L = lib()
    if L.zjni_isError(r):
        raise ZstdException(r)


def init(device_index=None):
    """Bind the calling thread to a GPU and create its persistent-kernel state."""
    if not torch.cuda.is_available():
        raise ZstdException(200, "zjni: no gfx950 device available")
    if device_index is None:
        device_index = torch.cuda.current_device()
    torch.cuda.set_device(device_index)
    r = lib().zjni_init(device_index)
    if r != 0:
        raise ZstdException(-r, "zjni_init failed")
    return device_index


def uniform_offsets(n, size, device):
    return torch.arange(0, n + 1, dtype=torch.int64, device=device) * size


def synth(n, buf_size, first_index=0, device="cuda"):
    """n mixed-entropy buffers of buf_size bytes generated in HBM."""
    blob = torch.empty(n * buf_size, dtype=torch.uint8, device=device)
    _check(lib().zjni_synth_fill_device(blob.data_ptr(), buf_size, first_index, n, _stream_ptr()))
    return blob


def compress(src_blob, src_off, dst_blob, dst_off, level=3, results=None, checksum=False, dictionary=None, hash_log=0, chain_log=0):
    """Enqueue zjni_compress_batch_device[2 / _usingCDict] on the current stream; returns the int64[n] result tensor
    (compressed size per buffer, or a negative ZSTD/ZJNI error code).  `dictionary`: a ZstdDictCompress (its level applies)."""
    n = src_off.numel() - 1
    if results is None:
        results = torch.empty(n, dtype=torch.int64, device=src_blob.device)
    if dictionary is not None:
        _check(lib().zjni_compress_batch_device_usingCDict(src_blob.data_ptr(), src_off.data_ptr(), dst_blob.data_ptr(), dst_off.data_ptr(),
                                                           results.data_ptr(), n, dictionary._ptr, 1 if checksum else 0, _stream_ptr()))
        return results
    if hash_log or chain_log:
        _check(lib().zjni_compress_batch_device_advanced(src_blob.data_ptr(), src_off.data_ptr(), dst_blob.data_ptr(), dst_off.data_ptr(),
                                                         results.data_ptr(), n, level, 1 if checksum else 0, hash_log, chain_log, _stream_ptr()))
        return results
    _check(lib().zjni_compress_batch_device2(src_blob.data_ptr(), src_off.data_ptr(), dst_blob.data_ptr(), dst_off.data_ptr(),
                                             results.data_ptr(), n, level, 1 if checksum else 0, _stream_ptr()))
    return results


def decompress(src_blob, src_off, dst_blob, dst_off, results=None, dictionary=None):
    """Enqueue zjni_decompress_batch_device[_usingDDict] on the current stream; returns int64[n] results."""
    n = src_off.numel() - 1
    if results is None:
        results = torch.empty(n, dtype=torch.int64, device=src_blob.device)
    dd = dictionary._ptr if dictionary is not None else None    # a zstd_jni_amd.ZstdDictDecompress
    _check(lib().zjni_decompress_batch_device_usingDDict(src_blob.data_ptr(), src_off.data_ptr(), dst_blob.data_ptr(), dst_off.data_ptr(),
                                                         results.data_ptr(), n, dd, _stream_ptr()))
    return results


def pack(results, dst_blob, dst_off, out=None, out_off=None):
    """Tightly pack a compress batch's variable-size outputs (sizes = results) into one blob:
    returns (packed_blob, packed_off int64[n+1]).  With `out` (uint8, capacity >= sum of sizes) and `out_off`
    (int64[n+1]) preallocated the exclusive scan and the byte movement are one library call
    (zjni_pack_batch_device2) and nothing synchronises with the host; without `out` the scan is torch's,
    because the blob's size has to come back first."""
    n = results.numel()
    if out_off is None:
        out_off = torch.zeros(n + 1, dtype=torch.int64, device=results.device)
    if out is not None and out_off.is_contiguous() and results.is_contiguous():
        if out_off.numel() != n + 1 or out_off.dtype != torch.int64 or out.dtype != torch.uint8 or results.dtype != torch.int64:
            raise ValueError("pack: out_off must be int64[n + 1], out uint8, results int64[n]")      # (the device writes n + 1 offsets: nothing else checks them)
        _check(lib().zjni_pack_batch_device2(dst_blob.data_ptr(), dst_off.data_ptr(), results.data_ptr(), out.data_ptr(),
                                             out_off.data_ptr(), n, _stream_ptr()))
        return out, out_off
    sizes = results.clamp(min=0)
    out_off[0] = 0
    torch.cumsum(sizes, 0, out=out_off[1:])
    if out is None:
        total = int(out_off[-1].item())
        out = torch.empty(max(total, 1), dtype=torch.uint8, device=results.device)[:total]
    _check(lib().zjni_pack_batch_device(dst_blob.data_ptr(), dst_off.data_ptr(), sizes.data_ptr(), out.data_ptr(),
                                        out_off.data_ptr(), n, _stream_ptr()))
    return out, out_off


def last_timing():
    """ms of the stages of the last large-batch device calls (HIP events on the launch stream, see zjni_last_timing2):
    {"match": .., "dec_prep": .., "dec_seq": .., "dec_exec": .., "dec_fused": .., "match_wide": ..}; -1 where a stage did not run."""
    import ctypes as C
    out = (C.c_float * 8)()
    _check(lib().zjni_last_timing2(out))
    return dict(zip(("match", "dec_prep", "dec_seq", "dec_exec", "dec_fused", "match_wide"), [float(x) for x in out][:6]))
