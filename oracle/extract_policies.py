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

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (fixture name, checkpoint, action box the checkpoint's JSON `data` member records)
#   forward/backward: tensegrity_env, 39-dim obs, trained with the older [-0.45, -0.15] box (README.md:30)
#   tracking: tr_env "tracking", 48-dim obs, trained with the current XML's [-0.45, 0.15] box
for name, path, low, high in (("forward", "best_models_pretrained/forward/SAC_5500000.zip", -0.45, -0.15),
                              ("backward", "best_models_pretrained/backward/SAC_4700000.zip", -0.45, -0.15),
                              ("tracking", "models_tracking/SAC_16525000.zip", -0.45, 0.15)):
    z = zipfile.ZipFile(os.path.join(REF, path))
    sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    arrs = {k.replace("actor.", "").replace(".", "_"): v.numpy().astype(np.float32) for k, v in sd.items()
            if k.startswith("actor.latent_pi") or k.startswith("actor.mu") or k.startswith("actor.log_std")}
    np.savez_compressed(os.path.join(OUT, f"policy_{name}.npz"), low=np.float32(low), high=np.float32(high), **arrs)
    print(name, {k: v.shape for k, v in arrs.items()})
