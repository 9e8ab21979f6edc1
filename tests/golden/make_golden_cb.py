"""Golden vectors of the reference's codebook training (src/quantization/cb_func.py), SURVEY 8(f) row 1.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_cb.py
Only outputs are stored (tests/golden/g7_cb_train.npz); inputs come from fpcodec_amd.synth.cb_training_vectors.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, synth  # noqa: E402


def main():
    import_reference()
    from quantization import cb_func
    data = synth.cb_training_vectors(3000)            # (3000, 17) float32, as train_cb.py:170-178 hands them over
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):             # update() prints per call
        # find_nearest / update / quantize on a fixed codebook
        cb0 = synth.cb_training_vectors(40, seed_offset=1).astype(np.float64)[:32]
        idx = cb_func.find_nearest(data, cb0)
        cb1 = cb_func.update(data, cb0.copy(), 32)
        qd = cb_func.quantize(cb1, data[:500])
        # empty cells: a codebook with far-away entries
        cb_far = cb0.copy()
        cb_far[5] += 100.0
        cb_far[17] -= 100.0
        cb2 = cb_func.update(data, cb_far, 32)
        # the whole splitting schedule, numpy's global RNG seeded
        np.random.seed(20221104)
        cbt = cb_func.vq_train(data, np.zeros((24, 17)), 24)
        # two-stage harvest as train_cb.py:186-193
        r = data.copy()
        stage = []
        np.random.seed(7)
        for i in range(2):
            c = cb_func.vq_train(r, np.zeros((8, 17)), 8)
            stage.append(c)
            r = cb_func.quantize(c, r) - r
        # the batch loop of train_cb.py:160-217 re-enacted with the reference's own cb_func on synthetic residual
        # batches (rows with all-zero rows mixed in, dropped as :186 drops them): first batch vq_train per stage,
        # second batch 10 x update per stage, unequal stage sizes
        n_entries = [12, 6]
        cbk10 = [np.zeros((n_entries[i], 17)) for i in range(2)]
        np.random.seed(31)
        errs = []
        for batch_idx in range(2):
            r10 = synth.cb_training_vectors(1500, seed_offset=50 + batch_idx)
            r10[::5] = 0.0
            r10 = np.array([r10[i] for i in range(len(r10)) if sum(abs(r10[i])) != 0])
            for i in range(2):
                if batch_idx == 0:
                    cbk10[i] = cb_func.vq_train(r10, cbk10[i], n_entries[i])
                else:
                    for _ in range(10):
                        cbk10[i] = cb_func.update(r10, cbk10[i], n_entries[i])
                qr = cb_func.quantize(cbk10[i], r10)
                r10 = qr - r10
            errs.append(float(np.sum(r10 * r10)))
    np.savez_compressed(os.path.join(HERE, "g10_train_cb_loop.npz"), stage0=cbk10[0], stage1=cbk10[1],
                        errs=np.array(errs), r_last=r10)
    np.savez_compressed(os.path.join(HERE, "g7_cb_train.npz"), idx=idx.astype(np.int64), cb1=cb1, qd=qd, cb2=cb2,
                        cbt=cbt, stage0=stage[0], stage1=stage[1], r_final=r)
    print("wrote g7_cb_train.npz", cbt.shape, idx[:8], float(np.sum(r * r)))


if __name__ == "__main__":
    main()
