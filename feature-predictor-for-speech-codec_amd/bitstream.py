"""Fixed-length bit packing of the encoder's symbols.

Per frame the encoder emits two threshold flags and up to three codebook indices
(`Wavernn.encoder(..., return_indices=True)`: {scalar idx, vq stage 1, vq stage 2, below-threshold vq idx},
-1 = not coded; the scalar idx is offset by the size of the above-threshold scalar codebook when it came
from the below-threshold one).  Frame layout, MSB first:

    ind1 (1 bit)  ind2 (1 bit)
    scalar index: ceil(log2 n_hi) bits if ind1 else ceil(log2 n_lo) bits if a below-threshold scalar codebook exists
    vector index: ceil(log2 N_hi0) [+ ceil(log2 N_hi1)] bits if ind2 else ceil(log2 N_lo) bits if a
                  below-threshold vector codebook exists

which is the fixed-length counterpart of the entropy figure `generate_qtz_features.py:94-101,202` reports
(entropy per used codebook x usage + 2 flag bits).  The pitch columns are side information and not part of
this stream.  Pure host code (numpy); the GPU work is on either side of it (`fpc_encode`,
`fpc_decode_features`)."""
import numpy as np


def _bits(n):
    return int(np.ceil(np.log2(n))) if n > 1 else 0


class Layout:
    """codebook sizes -> field widths.  sizes = (n_hi, n_lo, N_hi0, N_hi1, N_lo) as `Codebooks.sizes`"""

    def __init__(self, sizes):
        self.n_hi, self.n_lo, self.N_hi0, self.N_hi1, self.N_lo = (int(x) for x in sizes)
        self.b_scl_hi, self.b_scl_lo = _bits(self.n_hi), _bits(self.n_lo)
        self.b_v0, self.b_v1, self.b_vlo = _bits(self.N_hi0), _bits(self.N_hi1), _bits(self.N_lo)

    def frame_bits(self, ind1, ind2):
        b = 2
        b += self.b_scl_hi if ind1 else (self.b_scl_lo if self.n_lo else 0)
        b += (self.b_v0 + (self.b_v1 if self.N_hi1 else 0)) if ind2 else (self.b_vlo if self.N_lo else 0)
        return b


def pack(idx, sizes):
    """idx (L, 4) int -> (bytes, number of bits) for one utterance"""
    lay = Layout(sizes)
    idx = np.asarray(idx, dtype=np.int64)
    bits = []

    def put(v, n):
        if n:
            assert 0 <= v < (1 << n), (v, n)
            bits.extend((v >> (n - 1 - k)) & 1 for k in range(n))

    for s, v0, v1, vl in idx:
        ind1 = 0 <= s < lay.n_hi
        ind2 = v0 >= 0
        put(int(ind1), 1)
        put(int(ind2), 1)
        if ind1:
            put(int(s), lay.b_scl_hi)
        elif lay.n_lo:
            assert s >= lay.n_hi, "below-threshold scalar symbol missing"
            put(int(s - lay.n_hi), lay.b_scl_lo)
        else:
            assert s < 0
        if ind2:
            put(int(v0), lay.b_v0)
            if lay.N_hi1:
                put(int(v1), lay.b_v1)
        elif lay.N_lo:
            assert vl >= 0, "below-threshold vector symbol missing"
            put(int(vl), lay.b_vlo)
        else:
            assert vl < 0
    nbits = len(bits)
    return np.packbits(np.array(bits, dtype=np.uint8)).tobytes(), nbits


def unpack(data, nframes, sizes):
    """inverse of pack: (L, 4) int32 in the encoder's convention"""
    lay = Layout(sizes)
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
    pos = 0

    def get(n):
        nonlocal pos
        v = 0
        for k in range(n):
            v = (v << 1) | int(bits[pos + k])
        pos += n
        return v

    out = np.full((nframes, 4), -1, np.int32)
    for i in range(nframes):
        ind1, ind2 = get(1), get(1)
        if ind1:
            out[i, 0] = get(lay.b_scl_hi)
        elif lay.n_lo:
            out[i, 0] = lay.n_hi + get(lay.b_scl_lo)
        if ind2:
            out[i, 1] = get(lay.b_v0)
            if lay.N_hi1:
                out[i, 2] = get(lay.b_v1)
        elif lay.N_lo:
            out[i, 3] = get(lay.b_vlo)
    return out


def bits_per_frame(idx, sizes):
    """mean fixed-length bits per frame of a batch of symbol arrays (..., 4)"""
    lay = Layout(sizes)
    idx = np.asarray(idx).reshape(-1, 4)
    ind1 = (idx[:, 0] >= 0) & (idx[:, 0] < lay.n_hi)
    ind2 = idx[:, 1] >= 0
    b = 2.0 + np.where(ind1, lay.b_scl_hi, lay.b_scl_lo if lay.n_lo else 0) + np.where(
        ind2, lay.b_v0 + (lay.b_v1 if lay.N_hi1 else 0), lay.b_vlo if lay.N_lo else 0)
    return float(b.mean())


# ---------------------------------------------------------------------------------------------
# Entropy-coded framing: a static arithmetic coder (Witten-Neal-Cleary, 32-bit integer range) over
# the same per-frame fields, driven by symbol counts -- the codebook-usage histograms `cb_tot` the
# encoder returns (wavernn.py:189) plus the two flag rates.  With the counts of the coded material
# itself the size lands on the entropy figure of generate_qtz_features.py:94-101,202.
# ---------------------------------------------------------------------------------------------
_TOP, _HALF, _QTR = (1 << 32) - 1, 1 << 31, 1 << 30
_MAXTOT = 1 << 16


class Model:
    """cumulative frequency table of one field; every symbol keeps a count >= 1 so it stays codable"""

    def __init__(self, counts):
        c = np.maximum(np.asarray(counts, dtype=np.float64), 0.0)
        n = c.size
        if c.sum() <= 0:
            c = np.ones(n)
        scaled = np.maximum(1, np.floor(c / c.sum() * (_MAXTOT - n)).astype(np.int64))
        self.cum = np.concatenate([[0], np.cumsum(scaled)]).astype(np.int64)
        self.total = int(self.cum[-1])


class Models:
    """field models from the encoder's statistics: cb_tot = [scl_hi, scl_lo, vq stage 1, vq stage 2, vq_lo]
    usage histograms (`Wavernn.encoder`), keep = (ind1 rate, ind2 rate)"""

    def __init__(self, sizes, cb_tot, keep):
        self.lay = Layout(sizes)
        n = (self.lay.n_hi, self.lay.n_lo, self.lay.N_hi0, self.lay.N_hi1, self.lay.N_lo)
        tabs = []
        for i in range(5):  # an absent codebook has no symbols and no model
            h = np.asarray(cb_tot[i], dtype=np.float64).ravel()
            tabs.append(Model(h if h.size == n[i] else np.zeros(n[i])) if n[i] else None)
        self.scl_hi, self.scl_lo, self.v0, self.v1, self.vlo = tabs
        self.f1 = Model([1.0 - keep[0], keep[0]])
        self.f2 = Model([1.0 - keep[1], keep[1]])


def _fields(idx, lay):
    """yield (model name, symbol) in stream order for one utterance"""
    for s, v0, v1, vl in np.asarray(idx, dtype=np.int64):
        ind1, ind2 = 0 <= s < lay.n_hi, v0 >= 0
        yield "f1", int(ind1)
        yield "f2", int(ind2)
        if ind1:
            yield "scl_hi", int(s)
        elif lay.n_lo:
            yield "scl_lo", int(s - lay.n_hi)
        if ind2:
            yield "v0", int(v0)
            if lay.N_hi1:
                yield "v1", int(v1)
        elif lay.N_lo:
            yield "vlo", int(vl)


def entropy_pack(idx, models):
    """idx (L, 4) -> (bytes, number of bits)"""
    low, high, pend, bits = 0, _TOP, 0, []

    def emit(b):
        nonlocal pend
        bits.append(b)
        bits.extend([1 - b] * pend)
        pend = 0

    for name, sym in _fields(idx, models.lay):
        m = getattr(models, name)
        rng = high - low + 1
        high = low + rng * int(m.cum[sym + 1]) // m.total - 1
        low = low + rng * int(m.cum[sym]) // m.total
        while True:
            if high < _HALF:
                emit(0)
            elif low >= _HALF:
                emit(1)
                low -= _HALF
                high -= _HALF
            elif low >= _QTR and high < _HALF + _QTR:
                pend += 1
                low -= _QTR
                high -= _QTR
            else:
                break
            low, high = low << 1, (high << 1) | 1
    pend += 1
    emit(0 if low < _QTR else 1)
    return np.packbits(np.array(bits, dtype=np.uint8)).tobytes(), len(bits)


def entropy_unpack(data, nframes, models):
    """inverse of entropy_pack: (L, 4) int32 in the encoder's convention"""
    lay = models.lay
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
    pos = 0

    def nxt():
        nonlocal pos
        b = int(bits[pos]) if pos < bits.size else 0
        pos += 1
        return b

    low, high, val = 0, _TOP, 0
    for _ in range(32):
        val = (val << 1) | nxt()

    def get(name):
        nonlocal low, high, val
        m = getattr(models, name)
        rng = high - low + 1
        target = ((val - low + 1) * m.total - 1) // rng
        sym = int(np.searchsorted(m.cum, target, side="right")) - 1
        high = low + rng * int(m.cum[sym + 1]) // m.total - 1
        low = low + rng * int(m.cum[sym]) // m.total
        while True:
            if high < _HALF:
                pass
            elif low >= _HALF:
                low -= _HALF
                high -= _HALF
                val -= _HALF
            elif low >= _QTR and high < _HALF + _QTR:
                low -= _QTR
                high -= _QTR
                val -= _QTR
            else:
                break
            low, high, val = low << 1, (high << 1) | 1, (val << 1) | nxt()
        return sym

    out = np.full((nframes, 4), -1, np.int32)
    for i in range(nframes):
        ind1, ind2 = get("f1"), get("f2")
        if ind1:
            out[i, 0] = get("scl_hi")
        elif lay.n_lo:
            out[i, 0] = lay.n_hi + get("scl_lo")
        if ind2:
            out[i, 1] = get("v0")
            if lay.N_hi1:
                out[i, 2] = get("v1")
        elif lay.N_lo:
            out[i, 3] = get("vlo")
    return out
