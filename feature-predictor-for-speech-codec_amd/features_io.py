"""Feature-file formats either side of the hot path.  Host code only.

* LPCNet `.f32` dumps: flat float32, 36 values per 10 ms frame (20 cepstral/pitch + 16 LPC),
  `data_preprocess/write_small_files.py:18-24`.
* training/synthesis windows: `(n, 19, 36)` = 15-frame hops with 2 frames of context on either side
  (`write_small_files.py:58-66`); the datasets use the 15 centre frames `[:, 2:-2, :]`
  (`datasets/dataset_syn.py:86-97`, `dataset_orig.py:93-95`).
* the synthesis driver takes the LAST `chunks` windows of an utterance, pitch columns from the quantised
  feature file, normalised by 24.1 (`dataset_syn.py:71-99`).
* waveform export with the reference's std / peak normalisation (`synthesis_qtz.py:39-50`), 16-bit PCM WAV.
"""
import wave

import numpy as np

NB_FEATURES = 36
CHUNK = 15        # feature_chunk_size
CONTEXT = 2       # frames of context either side of a window
MAXI = 24.1       # synthesis_qtz.py:37, dataset_syn.py:43


def f32_to_windows(features):
    """flat float32 frames (or a path to a `.f32` dump) -> (n, 19, 36) overlapping windows, hop 15.
    `n = len // (15*36)` as `write_small_files.py:56`; the reference's strided view lets the last windows
    run past the data, here only windows that fit are returned."""
    a = np.fromfile(features, dtype=np.float32) if isinstance(features, str) else np.asarray(features, np.float32).ravel()
    nframes = a.size // NB_FEATURES
    a = a[: nframes * NB_FEATURES].reshape(nframes, NB_FEATURES)
    n = a.size // (CHUNK * NB_FEATURES)
    n = min(n, (nframes - (CHUNK + 2 * CONTEXT)) // CHUNK + 1) if nframes >= CHUNK + 2 * CONTEXT else 0
    if n <= 0:
        return np.zeros((0, CHUNK + 2 * CONTEXT, NB_FEATURES), np.float32)
    return np.stack([a[k * CHUNK: k * CHUNK + CHUNK + 2 * CONTEXT] for k in range(n)])


def frames_to_windows(frames):
    """encoded frames (L, 36) or (1, L, 36) -> (n, 19, 36) windows, hop 15, n = (L - 4) // 15: the as_strided view of
    src/generate_qtz_features.py:65-70 with its hard-coded (10, 19, 36) (right for that script's L = 154 only, SURVEY
    App. C) derived from L; frames that do not fill a whole window are dropped"""
    a = np.asarray(frames, dtype=np.float32).reshape(-1, NB_FEATURES)
    n = (len(a) - 2 * CONTEXT) // CHUNK
    if n <= 0:
        return np.zeros((0, CHUNK + 2 * CONTEXT, NB_FEATURES), np.float32)
    return np.stack([a[k * CHUNK: k * CHUNK + CHUNK + 2 * CONTEXT] for k in range(n)])


def save_windows(path, frames):
    """the per-utterance `{name}_features` file of the training / synthesis datasets (`write_small_files.py:66-70`
    stores the (n, 19, 36) windows with torch.save; the datasets read them back with torch.load)"""
    import torch
    w = frames if getattr(frames, "ndim", 0) == 3 and frames.shape[1:] == (CHUNK + 2 * CONTEXT, NB_FEATURES) \
        else frames_to_windows(frames)
    torch.save(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)), path)
    return w


def synthesis_frames(windows, qtz_windows=None, chunks=0):
    """`Libri_lpc_data_syn.__getitem__` (`dataset_syn.py:66-99`) for one utterance: the last `chunks`
    windows (all if 0; the utterance is doubled until it is long enough), centre frames only, pitch
    columns taken from the quantised file -> (nm_feat (chunks*15, 36) = feat / 24.1, qtz_feat)."""
    f = np.array(windows, dtype=np.float32, copy=True)
    q = np.array(qtz_windows if qtz_windows is not None else windows, dtype=np.float32, copy=True)
    n = min(len(f), len(q))
    f, q = f[:n], q[:n]
    f[:, :, -2:] = q[:, :, -2:]
    if chunks == 0:
        chunks = n
    while n < chunks:
        f, q = np.vstack((f, f)), np.vstack((q, q))
        n *= 2
    i = n - chunks if n > chunks else 0
    feat = f[i:i + chunks, CONTEXT:-CONTEXT, :].reshape(chunks * CHUNK, -1)
    qf = q[i:i + chunks, CONTEXT:-CONTEXT, :].reshape(chunks * CHUNK, -1)
    return feat / np.float32(MAXI), qf


def frames_to_f32(path, frames):
    """(L, 36) frames -> raw `.f32` file the vocoder CLI reads (`README.md:47`)"""
    np.asarray(frames, dtype=np.float32).reshape(-1, NB_FEATURES).tofile(path)


def normalise_wave(wave_in):
    """`saveaudio` (`synthesis_qtz.py:39-50`): flatten, divide by the standard deviation, then by the peak"""
    out = np.asarray(wave_in, dtype=np.float64).flatten().copy()
    out /= np.std(out)
    out /= max(abs(out))
    return out


def write_wav(path, wave_in, sr=16000, normalise=True):
    """16-bit PCM WAV (`sf.write(..., 16000, 'PCM_16')`): full scale 1.0 -> 32767 (libsndfile's float
    conversion: scale by 0x7FFF, round to nearest)"""
    x = normalise_wave(wave_in) if normalise else np.asarray(wave_in, dtype=np.float64).flatten()
    pcm = np.clip(np.rint(x * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())
    return pcm
