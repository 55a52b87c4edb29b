"""Extract the actor MLP of two shipped SB3 checkpoints into small .npz fixtures (tests/golden/policy_*.npz).

Runs only in the build container.  The zips are read with loaders that execute nothing from the file:
zipfile + torch.load(weights_only=True) for policy.pth.  Only tensors (weights) are kept; they are data, not code.
Used by tests/test_oracle_anchors.py::test_mujoco_trained_policies_move_the_right_way -- a behavioural anchor:
policies trained against real MuJoCo must walk in their intended direction in this simulator."""
import io
import os
import zipfile

import numpy as np
import torch

REF = "/root/reference/best_models_pretrained"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

for name, path in (("forward", "forward/SAC_5500000.zip"), ("backward", "backward/SAC_4700000.zip")):
    z = zipfile.ZipFile(os.path.join(REF, path))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    arrs = {k.replace("actor.", "").replace(".", "_"): v.numpy().astype(np.float32) for k, v in sd.items()
            if k.startswith("actor.latent_pi") or k.startswith("actor.mu")}
    # action box the checkpoint was trained with (README.md:30 / SURVEY.md App. C.8): [-0.45, -0.15]
    np.savez_compressed(os.path.join(OUT, f"policy_{name}.npz"), low=np.float32(-0.45), high=np.float32(-0.15), **arrs)
    print(name, {k: v.shape for k, v in arrs.items()})
