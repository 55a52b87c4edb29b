#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched tensegrity step on N MI355X (BASELINE.json metric).

One "step" = one control period (20 physics sub-steps of 1 ms) of every env: ONE pass of the hot path over the batch.
Workload at every N: BASELINE.json configs[1] per GPU -- 3-bar flat ground, tensegrity_env-v0 straight/+1, 4096 envs per
GPU, i.i.d. U[-0.45, 0.15]^6 float32 actions generated on the device before the timed region, auto-reset on (finished envs
restart from donor lanes inside the same launches, so resets are part of the measured time).  Weak scaling: rank g owns
global env ids [g*4096, (g+1)*4096); no collective on the data path, one RCCL all-gather of the rollout buffers
(obs, reward, done) per --gather-every steps, inside the timed region.

Prints ONE JSON line (rank 0).  roofline: algorithmic HBM bytes (674 B per env-step) over the step kernel's
mean duration measured with HIP events on the launch stream.  cpu_baseline: the CPU oracle (oracle/, fp64 restatement --
MuJoCo is not installed) timed on the host cores for a bounded sample; a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE = os.path.join(ROOT, "profiles", "round4_bench_rocprofv3_summary.json")
CENSUS = os.path.join(ROOT, "profiles", "round4_lane_census.json")


def profiled():
    """Counters of the dominant kernel from the committed rocprofv3 passes of this same command (tests/prof_bench.sh: FETCH_SIZE and
    WRITE_SIZE in separate passes, KB per dispatch, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note; SQ_INSTS_VALU).  They are
    NOT measured by this run: the line says where they come from, and they are dropped (null) when the kernel sources have changed
    since the profile was taken (hash of csrc/ + build flags stored with the profile)."""
    out = {"traffic": None, "valu_per_launch": None, "source": None, "launches": None}
    try:
        from tensegrity_rl_amd import build as _b
        d = json.load(open(PROFILE))
        if d.get("src_hash") != _b._src_hash():
            out["source"] = f"{os.path.relpath(PROFILE, ROOT)} is stale (kernel sources changed since): dropped"
            return out
        kn = d["dominant_kernel"]
        fe, wr, sq = d["fetch"][kn], d["write"][kn], d["sq"][kn]
        # with the donor stream a step is TWO dispatches of this kernel (live rows, donor rows): the counters are summed over both, per step
        per_step = 2 if len(d.get("step_kernel_row_by_grid_threads", {})) > 1 else 1
        out["traffic"] = (2.0 * fe["FETCH_SIZE"] / fe["dispatches"] + wr["WRITE_SIZE"] / wr["dispatches"]) * 1024.0 * per_step
        out["valu_per_launch"] = sq["SQ_INSTS_VALU"] / sq["dispatches"] * per_step
        out["launches"] = d.get("step_kernel_row_by_grid_threads")
        out["source"] = (f"{os.path.relpath(PROFILE, ROOT)} ({kn}, src_hash {d['src_hash'][:12]}; per step = {per_step} dispatch(es) of the kernel: "
                         "live rows on the caller's stream + donor rows on the donor stream)")
    except Exception as e:
        out["source"] = f"no profile ({type(e).__name__})"
    return out


BYTES_PER_ENV_STEP = 674  # fp32: 61 state words in + 6 action words in, 61 state words + 39 obs words + reward + 2 flags out
HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate 6290


def useful_lane_fraction():
    """Share of the lanes of a VALU instruction of the step kernel that own distinct work (the rest replicate a value or idle in their role):
    from the operation census of the host emulation of the same kernel source (tests/emul_census.py -> profiles/round4_lane_census.json)."""
    try:
        return float(json.load(open(CENSUS))["useful_lane_frac"])
    except Exception:
        return None


def vecenv_rate(n, env_id, device, steps=150, info_mode="minimal"):
    """env-steps/s through the numpy-facing SB3 surface -- TensegrityVecEnv.step(actions): host actions in, host obs / rewards / dones /
    infos out (what SB3's collect_rollouts drives, run.py:41-101).  info_mode="minimal": what SB3 itself reads; "full": the reference's info
    dict for every env (lazy snapshot mappings), with env 0's entries read every step the way run.py:139-161 reads them.  Measured on its
    own env AFTER the headline region; reported as config.vecenv_step_wait[_full]_env_steps_per_s, never as `value`."""
    import numpy as np
    from tensegrity_rl_amd import TensegrityVecEnv
    env = TensegrityVecEnv(n, env_id, device=device, desired_action="straight", desired_direction=1, seed=1234, n_donors=5 * n // 4, info_mode=info_mode)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32) for _ in range(8)]
    keys = ("x_position", "y_position", "x_velocity", "reward_forward") + (("tendon_length", "real_observation") if env_id.startswith("tr_env") else ())
    sink = 0.0
    for i in range(20):
        env.step(acts[i % 8])
    t0 = time.perf_counter()
    for i in range(steps):
        obs, rew, dones, infos = env.step(acts[i % 8])
        if info_mode == "full":
            i0 = infos[0]
            for k in keys:
                sink += float(np.sum(i0[k]))
    dt = time.perf_counter() - t0
    env.close()
    return n * steps / dt


def cpu_baseline(seconds, envs):
    """oracle batch (OpenMP over envs) on this host: env-steps/s with the same semantics (auto-reset included)."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 16))  # the GPU box grants a 16-core share per GPU
    try:  # the real engine when the box has it (it does not this round), else the CPU restatement below
        from oracle import mujoco_check
        if mujoco_check.available():
            import mujoco
            rate = mujoco_check.mujoco_rate(seconds, threads)
            return {"value": rate, "unit": "env-steps/s", "cores": threads, "kind": "reference",
                    "sample": f"mujoco {mujoco.__version__}, MJCF generated from the constants table (oracle/mjcf.py), {threads} processes x "
                              f"{seconds:.0f} s of mj_step(nstep=20) + mj_rnePostConstraint with random actions"}
    except Exception:
        pass
    envs = 4 * threads
    ob = O.OracleBatch(O.make_cfg("tensegrity_env", "straight"), envs, seed=1234, threads=threads)
    ob.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-0.45, 0.15, (8, envs, 6)).astype(np.float32)
    ob.step(acts[0])
    t0 = time.time(); steps = 0
    while time.time() - t0 < seconds:
        ob.step(acts[steps % 8]); steps += 1
    dt = time.time() - t0
    ob.close()
    return {"value": envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{envs} envs x {steps} steps ({dt:.1f} s), oracle/ fp64 CPU restatement with OpenMP over envs, auto-reset on"}


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` as a bare command: start N fresh rank processes through torch.distributed.run (one per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, forward rank 0's JSON line, return the launcher's exit code.
    This parent never initialises HIP (and never execs): it only waits for the children."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, TGR_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        try:
            d = json.loads(ln)
            if isinstance(d, dict) and "metric" in d:
                line = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n_ranks}-rank launch failed (exit code {proc.returncode}, JSON line {'found' if line else 'missing'})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def dry_run(args, rank, world):
    """TGR_BENCH_DRYRUN=1: launcher / rendezvous / timing-protocol plumbing only, on CPU over gloo -- no env is built and nothing is
    measured (`value` is null); used by tests/test_bench_launcher.py where there is no GPU."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        got = [None] * world
        dist.all_gather_object(got, rank)
        assert got == list(range(world))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec (whole node)", "value": None, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--env-id", default="tensegrity_env-v0")
    ap.add_argument("--model", default="flat", choices=["flat", "uneven"],
                    help="uneven: BASELINE config #4 (heightfield floor, tests/golden/terrain_100.npy); not the headline metric")
    ap.add_argument("--donors", type=int, default=None,
                    help="donor lanes per GPU (episodes being settled for the next resets); default 5/4 of the envs: an env lives ~47 steps "
                         "under random actions and reset_model() costs 50, so ~1.06 N lanes are busy settling at any time")
    ap.add_argument("--kernel-layout", default="auto", choices=["auto", "row", "block"])
    ap.add_argument("--gather-every", type=int, default=16)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vecenv", action="store_true", help="skip the numpy-facing VecEnv.step() measurement (reported beside the headline)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("TGR_BENCH_DRYRUN") == "1":
        return dry_run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist
    from tensegrity_rl_amd import TensegrityVecEnv
    from tensegrity_rl_amd.sharding import RolloutGatherer, shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: under torch.distributed.run use --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` bare: it starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the HIP path is the product and there is no CPU fallback")
    # TGR_BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo -- exercises the N>1 code path on a one-GPU box
    # (numbers from it mean nothing); the real thing is one rank per GPU over RCCL
    rehearsal = os.environ.get("TGR_BENCH_REHEARSAL") == "1"
    local_rank = 0 if rehearsal else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TGR_BENCH_FORCE_DIST=1: run the N>1 code path (process group over RCCL, all-gathers, max-reduce of the time) with a single
    # rank -- checks that RCCL comes up on a one-GPU box; costs a little, so it is never the default
    force_dist = os.environ.get("TGR_BENCH_FORCE_DIST") == "1" and world == 1
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29531"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n_local = args.envs_per_gpu
    n_total = n_local * world
    lo, hi = shard_range(rank, world, n_total)
    extra = {}
    if args.model == "uneven":
        import numpy as np
        extra = dict(model="uneven", terrain=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "terrain_100.npy")))
    donors = args.donors if args.donors is not None else (5 * n_local // 4 if args.kernel_layout != "block" else 2 * n_local)
    env = TensegrityVecEnv(n_local, args.env_id, device=local_rank, desired_action="straight", desired_direction=1,
                           seed=1234, first_env_id=lo, n_donors=donors, kernel_layout=args.kernel_layout, **extra)
    row_layout = args.kernel_layout != "block"  # (the row kernels serve both models)
    env.reset_tensors()
    g = torch.Generator(device=dev); g.manual_seed(rank)
    acts = [(torch.rand((n_local, 6), device=dev, generator=g) * 0.6 - 0.45).contiguous() for _ in range(32)]
    gather = (RolloutGatherer(n_local, env.obs_dim, args.gather_every, dev, world_size=world, always_collective=force_dist, overlap=True)
              if (world > 1 or force_dist) else None)

    def one_step(i):
        obs, rew, term, trunc, _, _ = env.step_tensors(acts[i % 32])
        if gather is not None:
            gather.push(obs, rew, term, trunc)

    for i in range(args.warmup):
        one_step(i)
    stats0 = env.stats()
    # Kernel-only timing of the dominant kernel -- the step kernel over the live rows -- with HIP events the library records around exactly that
    # launch, on the stream it is launched on, INSIDE the timed region (roofline.achieved) -- on one step in four: an event pair is two packets on
    # the stream, and timed on every step the run is 2 % slower for its own measurement.  The same steps carry the torch events of step_chain_ms.
    every = 4
    n_timed = (args.steps + every - 1) // every
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_timed)]
    env.time_kernels(n_timed, every)
    multi = world > 1 or force_dist
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % every == 0:
            evs[i // every][0].record()
        one_step(args.warmup + i)
        if i % every == 0:
            evs[i // every][1].record()
    if gather is not None:
        gather.flush()  # the last chunk's all-gather belongs to the timed region
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats1 = env.stats()
    step_ms = sorted(a.elapsed_time(b) for a, b in evs)
    chain_ms = sum(step_ms) / len(step_ms)   # one step's work on the caller's stream: step kernel over the live rows + match + settle + gaps
    kt = sorted(env.kernel_times_ms(n_timed))
    env.time_kernels(0)
    kern_ms = sum(kt) / len(kt) if kt else chain_ms

    if rank == 0:
        value = n_total * args.steps / dt
        launch_bytes = BYTES_PER_ENV_STEP * n_local
        achieved = launch_bytes / (kern_ms * 1e-3) / 1e9
        resets = stats1["donor_resets"] - stats0["donor_resets"] + stats1["settle_kernel_resets"] - stats0["settle_kernel_resets"]
        # the committed counters belong to the default command: attach them to that workload only
        default_workload = args.model == "flat" and args.env_id == "tensegrity_env-v0" and n_local == 4096 and row_layout and args.donors is None
        prof = profiled() if default_workload else {"traffic": None, "valu_per_launch": None, "launches": None, "source": "not the profiled (default) workload"}
        lane_sub = max(stats1["lane_substeps"] - stats0["lane_substeps"], 1)
        ulf = useful_lane_fraction()
        # (the profile's VALU count is per STEP -- both launches of the kernel, which overlap -- so the rate is over the step's wall time)
        valu_rate = (prof["valu_per_launch"] * 64 / (dt / args.steps)) if prof["valu_per_launch"] else None
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3-bar {'uneven-ground (heightfield)' if args.model == 'uneven' else 'flat-ground'} {args.env_id} straight/+1, N_envs={n_local} per GPU ({n_total} total), "
                                   "random-action rollout U[-0.45,0.15]^6, auto-reset via donor lanes, 20 sub-steps of 1 ms per step",
                       "envs_per_gpu": n_local, "donor_lanes_per_gpu": donors,
                       "kernel_layout": "row (one env per 16-lane DPP row; fp32 with pose / kinematics / tendon geometry / axial torques in fp64; qpos kept in fp64 in HBM)" if row_layout
                                        else "block (64 envs x 4 wavefronts, LDS columns; fp32 throughout)",
                       "gather_every": args.gather_every if (world > 1 or force_dist) else None,
                       "collective_backend": (dist.get_backend() if (world > 1 or force_dist) else None),
                       "all_gathers_issued": (gather.collectives if gather is not None else 0),
                       "resets_per_step_rank0": resets / args.steps,
                       "newton_iters_per_substep": stats1["newton_iters_per_substep"],
                       "unconverged_lane_substeps_per_env_step": (stats1["unconverged_lane_substeps"] - stats0["unconverged_lane_substeps"]) / (n_local * args.steps),
                       "unconverged_fraction_of_lane_substeps": (stats1["unconverged_lane_substeps"] - stats0["unconverged_lane_substeps"]) / lane_sub,
                       "dropped_contacts_per_env_step": (stats1["dropped_contacts"] - stats0["dropped_contacts"]) / (n_local * args.steps),
                       "starved_resets_rank0": stats1["settle_kernel_resets"] - stats0["settle_kernel_resets"],
                       # a donor lane needs 51 launches to settle its first episode: a timed window that starts before ~100 steps sees the
                       # pool's start-up transient (every donor wavefront busy with one row, more Newton iterations per sub-step)
                       "donor_pool_warm": bool(args.warmup >= 100),
                       # what "1e-4 relative" means wherever this repo quotes it (tests/gpu_common.py rel_env): per env-step and per env,
                       # max_i |err_i| / max(max_i |ref_i|, floor) over the 21 qpos (floor 1e-3) or the 18 qvel (floor 1e-2) components
                       "parity_tolerance": "per env-step: |err|_inf / max(|ref|_inf, floor), floor 1e-3 (qpos) / 1e-2 (qvel); >= 99.9 % of env-steps "
                                           "inside 1e-4, the rest classified contact flips (tests/test_gpu_parity.py, tests/test_gpu_uneven.py)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": prof["traffic"], "traffic_source": prof["source"],
                         "note": "step kernel: 674 algorithmic B per env-step x envs per launch / mean duration of the launch that steps the live "
                                 "rows (HIP events the library records around exactly that launch, on the caller's stream, inside the timed region, one step in "
                                 "four: tgr_time_kernels_every; "
                                 "step_chain_ms = that launch + match + settle launches + gaps, from events around tgr_step; the donor rows run "
                                 "concurrently in their own launch on the handle's low-priority stream); the path is VALU-issue bound (~1e6 flop "
                                 "per env-step), not HBM bound: see DESIGN.md",
                         "kernel_ms": kern_ms, "kernel_ms_p50": kt[len(kt) // 2] if kt else None,
                         "step_chain_ms": chain_ms,
                         # rocprofv3 --kernel-trace of the same command: the two dispatches of the kernel per step, by grid size (threads)
                         "profiled_launches_by_grid_threads": prof["launches"],
                         # the practical bound is VALU issue of the resident waves, so report those too
                         "resident_waves": ((n_local + donors + 3) // 4) if row_layout else 4 * ((n_local + donors + 63) // 64),
                         # 78.6e12 = 1024 SIMDs x 32 lanes x 2.4 GHz: one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md)
                         "valu_lane_ops_per_s": valu_rate,
                         "valu_frac_of_peak": (valu_rate / 78.6e12) if valu_rate else None,
                         # the same rate counting only lanes that own distinct work (replicated values and idle role lanes excluded)
                         "useful_lane_frac": ulf,
                         "valu_useful_frac_of_peak": (valu_rate * ulf / 78.6e12) if (valu_rate and ulf) else None},
        }
        if not args.no_vecenv and world == 1 and args.model == "flat":
            for key, mode in (("vecenv_step_wait_env_steps_per_s", "minimal"), ("vecenv_step_wait_full_env_steps_per_s", "full")):
                try:
                    out["config"][key] = vecenv_rate(n_local, args.env_id, local_rank, info_mode=mode)
                except Exception as e:
                    out["config"][key] = f"failed: {e}"
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed once: rank 0 of the one-GPU run
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, 64)
            except Exception as e:  # the oracle is test infrastructure; never let it break the bench line
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    env.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
