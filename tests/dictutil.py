"""Hand-assembled zstd dictionaries for tests: entropy sections the trainer never produces (tables that do not
cover every symbol -> the encoder's "check" repeat modes, short dictIDs).  Format: magic 0xEC30A437, dictID,
Huffman weights, OF / ML / LL NCounts, three repcodes, content (reference N/compress/zstd_compress.c:5061-5155)."""
import heapq
import struct


def write_ncount(norm, table_log):
    """NCount header for a normalised distribution (sum of |norm| == 1 << table_log); -1 = low-probability symbol."""
    out = bytearray()
    table_size = 1 << table_log
    remaining, threshold, nb_bits = table_size + 1, table_size, table_log + 1
    bit_stream, bit_count = table_log - 5, 4
    symbol, alphabet, previous_is0 = 0, len(norm), False
    while symbol < alphabet and remaining > 1:
        if previous_is0:
            start = symbol
            while symbol < alphabet and norm[symbol] == 0:
                symbol += 1
            if symbol == alphabet:
                break
            while symbol >= start + 24:
                start += 24
                bit_stream += 0xFFFF << bit_count
                out += struct.pack("<H", bit_stream & 0xFFFF)
                bit_stream >>= 16
            while symbol >= start + 3:
                start += 3
                bit_stream += 3 << bit_count
                bit_count += 2
            bit_stream += (symbol - start) << bit_count
            bit_count += 2
            if bit_count > 16:
                out += struct.pack("<H", bit_stream & 0xFFFF)
                bit_stream >>= 16
                bit_count -= 16
        count = norm[symbol]
        symbol += 1
        mx = (2 * threshold - 1) - remaining
        remaining -= abs(count)
        count += 1
        if count >= threshold:
            count += mx
        bit_stream += count << bit_count
        bit_count += nb_bits
        bit_count -= 1 if count < mx else 0
        previous_is0 = count == 1
        assert remaining >= 1
        while remaining < threshold:
            nb_bits -= 1
            threshold >>= 1
        if bit_count > 16:
            out += struct.pack("<H", bit_stream & 0xFFFF)
            bit_stream >>= 16
            bit_count -= 16
    assert remaining == 1
    out += struct.pack("<H", bit_stream & 0xFFFF)
    return bytes(out[: len(out) - 2 + (bit_count + 7) // 8])


def normalise(weights, table_log):
    """integer weights (0 = absent) -> counts summing to 1 << table_log, every present symbol >= 1"""
    total = sum(weights)
    size = 1 << table_log
    norm = [max(1, w * size // total) if w else 0 for w in weights]
    diff = size - sum(norm)
    big = max(range(len(norm)), key=lambda i: norm[i])
    norm[big] += diff
    assert norm[big] >= 1 and sum(norm) == size
    return norm


def huffman_weights(hist, max_bits=11):
    """direct (4-bit) Huffman weight header for a byte histogram whose symbols are all < 129"""
    hist = list(hist)
    while True:
        heap = [(c, i, None, None) for i, c in enumerate(hist) if c]
        heapq.heapify(heap)
        uid = 1000
        while len(heap) > 1:
            a = heapq.heappop(heap)
            b = heapq.heappop(heap)
            heapq.heappush(heap, (a[0] + b[0], uid, a, b))
            uid += 1
        lengths = {}

        def walk(n, d):
            if n[2] is None:
                lengths[n[1]] = max(d, 1)
            else:
                walk(n[2], d + 1)
                walk(n[3], d + 1)

        walk(heap[0], 0)
        if max(lengths.values()) <= max_bits:
            break
        hist = [(c + 1) // 2 + 1 if c else 0 for c in hist]      # flatten and retry
    top = max(lengths.values())
    last = max(lengths)
    assert last <= 128
    w = [(top + 1 - lengths[s]) if s in lengths else 0 for s in range(last)]      # the last symbol's weight is implied
    if len(w) % 2:
        w.append(0)
    body = bytes((w[i] << 4) | w[i + 1] for i in range(0, len(w), 2))
    return bytes([127 + last]) + body


def build(content, dict_id, lit_hist, of_norm, of_log, ml_norm, ml_log, ll_norm, ll_log, reps=(1, 4, 8), huf_max_bits=11):
    return (struct.pack("<II", 0xEC30A437, dict_id) + huffman_weights(lit_hist, huf_max_bits) + write_ncount(of_norm, of_log)
            + write_ncount(ml_norm, ml_log) + write_ncount(ll_norm, ll_log) + struct.pack("<III", *reps) + content)
