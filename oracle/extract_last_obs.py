"""Extract the observations recorded inside the reference's SB3 checkpoints into one small fixture (tests/golden/mujoco_last_obs.json).

Every SB3 zip keeps the last observation its (MuJoCo-backed) env returned -- `_last_obs` / `_last_original_obs` in the JSON member `data`
(SURVEY.md App. D): real MuJoCo output for one mid-episode state per checkpoint.  tensegrity_env (39): three geom quaternions, qvel, nine
tendon lengths; tr_env (27 / 45 / 48): six cap positions about their centroid, (cap velocities,) nine tendon lengths (, 3 command values).
Positions AND tendon lengths of the same state over-determine the bars' relative pose, so each record is an exact check of the site /
tendon table and the frame conventions against MuJoCo-written numbers (tests/test_oracle_anchors.py).

Runs only in the build container.  The files are read with loaders that execute nothing from them: zipfile + json for the member, and
the base64 field -- a pickled numpy array -- is NOT unpickled: pickletools.genops only disassembles the opcode stream, and the array's raw
buffer (the one bytes / bytearray argument of 8 * D bytes) is taken from it.  Only numbers are kept; they are data, not code."""
import base64
import json
import os
import pickletools
import zipfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mujoco_last_obs.json")


def raw_array(b64, dim):
    """the float64 buffer of a pickled ndarray, found by disassembly (nothing is executed or constructed)"""
    blob = base64.b64decode(b64)
    hits = [arg for op, arg, _ in pickletools.genops(blob) if isinstance(arg, (bytes, bytearray)) and len(arg) == 8 * dim]
    if len(hits) != 1:
        return None
    return np.frombuffer(bytes(hits[0]), dtype="<f8")


def main():
    out = []
    for root, _, files in sorted(os.walk(REF)):
        for f in sorted(files):
            if not f.endswith(".zip"):
                continue
            path = os.path.join(root, f)
            try:
                d = json.loads(zipfile.ZipFile(path).read("data"))
            except Exception as e:  # not an SB3 archive
                print("skipped", path, e); continue
            shape = d.get("observation_space", {}).get("_shape")
            if not shape or len(shape) != 1:
                continue
            dim = int(shape[0])
            rec = {"file": os.path.relpath(path, REF), "obs_dim": dim, "n_envs": d.get("n_envs"),
                   "num_timesteps": d.get("num_timesteps"), "action_low": d.get("action_space", {}).get("low_repr")}
            for key in ("_last_obs", "_last_original_obs"):
                v = d.get(key)
                a = raw_array(v[":serialized:"], dim) if isinstance(v, dict) and ":serialized:" in v else None
                rec[key] = None if a is None else [float(x) for x in a]
            out.append(rec)
            print(rec["file"], dim, "last_obs" if rec["_last_obs"] else "-", "original" if rec["_last_original_obs"] else "-")
    with open(OUT, "w") as fh:
        json.dump({"source": "SB3 checkpoints of the reference (data JSON, _last_obs / _last_original_obs); see oracle/extract_last_obs.py",
                   "records": out}, fh, indent=0)
    print(len(out), "records ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
