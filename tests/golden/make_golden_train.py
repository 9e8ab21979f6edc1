"""Golden vectors of the reference's live predictor-training step (src/train_frame.py:53-120, batch_idx <= 10
branch: teacher-forced forward, nn.MSELoss against the next frame, Adam(lr=1e-4)) -- SURVEY 8(f) row 4.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_train.py
The reference's own Wavernn class, torch autograd and torch.optim.Adam do the work; only outputs are stored
(loss per step, every STRIDE-th element of each gradient of step 1 and of each parameter after step 2)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, synth  # noqa: E402

STRIDE = 37
KEYS = ["rnn1.weight_ih_l0", "rnn1.weight_hh_l0", "rnn1.bias_ih_l0", "rnn1.bias_hh_l0", "rnn2.weight_ih_l0",
        "rnn2.weight_hh_l0", "rnn2.bias_ih_l0", "rnn2.bias_hh_l0", "dual_fc.0.weight", "dual_fc.0.bias"]


def main():
    torch.set_num_threads(1)
    wavernn = import_reference()[0]
    model = wavernn.Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.predictor_state_dict().items()})
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)          # train_frame.py:250
    mseloss = torch.nn.MSELoss()                                  # :39
    feat = torch.from_numpy(synth.predictor_features(6, 40, utt0=4000))
    params = dict(model.named_parameters())
    out = {}
    for step in range(2):
        feat_out, _, _ = model(feat)                              # :78
        loss = mseloss(feat_out[:, :-1, :], feat[:, 1:, :18])     # :79
        opt.zero_grad()
        loss.backward()
        if step == 0:
            for k in KEYS:
                g = params[k].grad.detach().numpy().ravel()
                out["g_" + k] = g[::STRIDE].copy()
                out["gn_" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        opt.step()
        out[f"loss{step}"] = np.float64(loss.item())
    for k in KEYS:
        out["p_" + k] = params[k].detach().numpy().ravel()[::STRIDE].copy()
    np.savez_compressed(os.path.join(HERE, "g8_train_step.npz"), **out)
    print("wrote g8_train_step.npz", out["loss0"], out["loss1"], {k: float(out["gn_" + k]) for k in KEYS[:3]})


if __name__ == "__main__":
    main()
