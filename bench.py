#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec for the 160-level Arbo SMT-verifier circuit
(Poseidon-BN254), batch 1024 per GPU, plus the MSM kernel's achieved algorithmic GB/s.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole prove path (witness solve -> quotient (6 NTTs) -> 4 G1 + 1 G2 MSM ->
assembly) over one batch of synthetic witnesses that is already resident in HBM.  Ranks are
independent (weak scaling: every rank proves its own batch; no data-path collective).
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def available_cpus():
    """Host threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    documented per-GPU CPU share (16) as a ceiling; ZKMI_CPU_THREADS overrides."""
    if os.environ.get("ZKMI_CPU_THREADS"):
        return int(os.environ["ZKMI_CPU_THREADS"])
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per GPU per step")
    ap.add_argument("--workload", default="arbo",
                    help="arbo (the headline, BASELINE config 2) | poseidon | verifier | "
                         "elgamal-add | elgamal-encrypt | address: the other configs, for "
                         "DESIGN.md's table (distinct witnesses are cycled to fill the batch)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct witnesses to generate (0: one per proof)")
    ap.add_argument("--levels", type=int, default=160)
    ap.add_argument("--populated", type=int, default=10, help="non-zero siblings per path")
    ap.add_argument("--window-g1", type=int, default=0)
    ap.add_argument("--window-g2", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="proofs for the CPU baseline (-1: 2 per core, 0: skip)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=-1,
                    help="rehearsal only: put every rank on this device instead of LOCAL_RANK")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one blocking zkmi_prove_batch per step (no overlap of consecutive steps)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gnark_crypto_primitives_amd import backend, groth16, lib, workloads
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array

    rank, world, local_rank = backend.env_rank_world()
    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # collectives' tensors
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)
    if world > 1:
        # first collective now: RCCL's buffers are allocated before the MSM tables size themselves
        # against the free HBM
        t = torch.zeros(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        if cdev.type == "cuda":
            torch.cuda.synchronize()
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if (args.verbose and rank == 0) \
        else (lambda *a: None)

    t0 = time.time()
    ctx = lib.Context(local_rank)
    circuit, gen, label = workloads.build(args.workload, args.levels, args.populated)
    cc = compile_circuit(circuit)
    log(f"compiled: {cc.n_constraints} constraints, {cc.n_wires} wires, {cc.n_ops} ops "
        f"({time.time() - t0:.1f}s)")
    pk, vk, _ = groth16.setup(cc, 2, groth16.gpu_mul(ctx))
    log(f"setup: log_n={pk.log_n} A={len(pk.a_wire)} B={len(pk.b_wire)} K={len(pk.k_wire)} "
        f"Z={pk.g1_z.shape[0]} ({time.time() - t0:.1f}s)")
    prover = groth16.Prover(ctx, cc, pk, args.window_g1, args.window_g2)
    log(f"key resident, window tables built ({time.time() - t0:.1f}s)")

    # synthetic witnesses, seeded per rank
    rng = random.Random(1000 + rank)
    B = args.batch
    n_distinct = min(B, args.distinct) if args.distinct > 0 else B
    ws = [to_mont_array(cc.assignment_vector(gen(rng))) for _ in range(n_distinct)]
    inp_h = np.stack([ws[i % n_distinct] for i in range(B)])
    rs_h = np.stack([to_mont_array([rng.randrange(workloads.R), rng.randrange(workloads.R)])
                     for _ in range(B)])
    inp_d = torch.from_numpy(inp_h.view(np.int64)).to(dev)
    rs_d = torch.from_numpy(rs_h.view(np.int64)).to(dev)
    proofs_d = torch.zeros((B, 32), dtype=torch.int64, device=dev)
    status_d = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    log(f"witnesses resident ({time.time() - t0:.1f}s)")

    # Steps are software-pipelined two deep through the library's submit/collect pair: the
    # latency-bound witness solve of step k+1 (16 wavefronts) runs on a second HIP stream under the
    # NTT/MSM kernels of step k.  The timed region contains K submits and K collects: the first
    # solve is exposed, nothing of the timed work happens outside the region.
    for _ in range(args.warmup):
        prover.prove(inp_d, rs_d, proofs_d, status_d)
    stage = np.zeros(8)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    if args.no_pipeline:
        for _ in range(args.steps):
            prover.prove(inp_d, rs_d, proofs_d, status_d)
            stage += np.array(ctx.last_timings())
    else:
        prover.submit(inp_d, rs_d)
        for k in range(args.steps):
            if k + 1 < args.steps:
                prover.submit(inp_d, rs_d)
            prover.collect(proofs_d, status_d)
            stage += np.array(ctx.last_timings())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = status_d.cpu().numpy()
    n_bad = int((status != 0).sum())
    if world > 1:
        t = torch.tensor([n_bad], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        n_bad = int(t.item())
    proofs = proofs_d.cpu().numpy().view(np.uint64)

    if rank == 0:
        stage /= max(args.steps, 1)
        # ---- roofline of the dominant kernel: the G1 accumulate kernel, four launches per step
        # (msm_accumulate_shared<Fq> under the default shared-table plan; stage[6] = sum of the HIP
        # event pairs that bracket each accumulate launch alone)
        ns = [len(pk.a_wire), len(pk.b_wire), len(pk.k_wire), pk.g1_z.shape[0]]
        alg_bytes = sum(n * 64 + B * n * 32 for n in ns)
        msm_s = stage[6] * 1e-3
        achieved = alg_bytes / msm_s / 1e9 if msm_s > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE +
        # WRITE_SIZE of the four launches, raw counter values; same circuit, batch and windows)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if pm.get("batch") == B and pm.get("levels") == args.levels and \
                    args.workload == "arbo":
                traffic = pm["msm_g1_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        info = ctx.pk_info(prover.pk_h)
        kname = ("msm_accumulate_comb<Fq, false>" if info["g1_comb_k"] else
                 "msm_accumulate_shared<Fq>" if info["g1_shared"] else "msm_accumulate<Fq, false>")
        roofline = {"bound": "hbm", "kernel": f"{kname} (G1 MSMs of the key)",
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": traffic,
                    "launches_per_step": 4, "avg_launch_ms": stage[6] / 4.0,
                    "algorithmic_bytes_per_launch": alg_bytes / 4.0}
        # The kernel is bound by the integer multiplier, not HBM: report the
        # v_mad_u64_u32 view beside the HBM one.  1467 mads per mixed addition (csrc/ec29.h: 6 products x 162, 2 squares x 126, one
        # two-product sum with a shared reduction 243), one
        # addition per (base, window, proof); peak 3.55e13 lane-mads/s measured by
        # tools/instr_rate.hip (profiles/r01_instr_rate.log).
        if info["g1_comb_k"]:      # one addition per (group of k bases, bit, proof)
            madds = sum(-(-n // info["g1_comb_k"]) for n in ns) * info["g1_windows"] * B
        else:
            madds = sum(ns) * info["g1_windows"] * B
        roofline["alu"] = {"unit": "v_mad_u64_u32 lane-ops/s", "achieved": madds * 1467 / msm_s,
                           "peak": 3.55e13, "frac": madds * 1467 / msm_s / 3.55e13}
        try:
            roofline["alu"]["fq_mul_per_s_ff29_microbench"] = ctx.field_mul_bench(2, 1 << 22, 256)
        except Exception:
            pass
        cpu = None
        if args.cpu_sample != 0 and world == 1:      # CPU baseline: rank 0 at N = 1 only
            from oracle import cref
            cores = available_cpus()
            S = args.cpu_sample if args.cpu_sample > 0 else min(B, 2 * cores)
            rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
            tc = time.perf_counter()
            want, wstatus, used = cref.groth16_prove_batch(rh, ph, inp_h[:S], rs_h[:S], cores)
            tc = time.perf_counter() - tc
            same = bool(np.array_equal(want, proofs[:S]) and not wstatus.any())
            cpu = {"value": S / tc, "unit": "proofs/s", "cores": used, "kind": "port",
                   "sample": f"first {S} proofs of the same batch, C oracle (oracle/c), "
                             f"OpenMP over proofs", "seconds": tc,
                   "gpu_proofs_bit_exact_vs_cpu": same}
        out = {
            "metric": "proofs/sec, Arbo-160 Poseidon SMT-verifier circuit, Groth16/BN254"
                      if args.workload == "arbo" and args.levels == 160
                      else f"proofs/sec, {label}, Groth16/BN254",
            "value": world * B * args.steps / elapsed, "unit": "proofs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 Montgomery (254-bit integer)",
            "data": "synthetic",
            "config": {"workload": f"{label}, batch {B} proofs per GPU",
                       "constraints": cc.n_constraints, "wires": cc.n_wires,
                       "domain_log2": pk.log_n, "batch_per_gpu": B,
                       "msm_terms_per_proof": {"g1": int(sum(ns)), "g2": ns[1]},
                       "msm_window_tables": ctx.pk_info(prover.pk_h),
                       "parallelism": f"batch-split x{world}, no collective"},
            "pipelined": not args.no_pipeline,
            "stage_ms": {"solve": stage[0], "quotient_ntt": stage[1], "msm_g1": stage[2],
                         "msm_g2": stage[3], "assemble_overlapped": stage[4], "main_stream_span": stage[5],
                         "msm_g1_kernel_only": stage[6], "msm_g2_kernel_only": stage[7]},
            "unsatisfied": n_bad,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    prover.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
