#!/usr/bin/env python3
"""Headline benchmark: Groth16 proofs/sec for the 160-level Arbo SMT-verifier circuit
(Poseidon-BN254), batch 1024 per GPU, plus the MSM kernel's achieved algorithmic GB/s.

  python bench.py --gpus N --steps K --warmup W
  N > 1 either way: `python bench.py --gpus N` starts N fresh rank processes by itself (one per GPU,
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's JSON line forwarded), and under
  `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one of the
  ranks.  --gpus that disagrees with an inherited WORLD_SIZE is an error.

One step = one pass of the whole prove path (witness solve -> quotient (6 NTTs) -> 4 G1 + 1 G2 MSM ->
assembly) over one batch of synthetic witnesses that is already resident in HBM, ending with the
RCCL all_gather of the 264-byte proof records (the only exchange the path has).  Default scaling is
weak (every rank proves its own batch of --batch proofs, the headline configuration);
--scaling strong splits ONE global batch of --batch proofs across the ranks (BASELINE configs 3 and
4: --workload verifier --batch 4096, --workload elgamal-add --batch 8192).
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


def host_cpus():
    """Like available_cpus() without the per-GPU ceiling: what the whole job may use."""
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
    return n


# ---- start-up cache: compiled circuit and synthetic witnesses on local disk, so that ranks
# 1..N-1 (and the next run on the same box: the driver runs N = 1, 2, 4, 8 back to back) load what
# one process already computed.  Files are this program's own output (pickle / .npy), written
# atomically; a flock per file makes exactly one process compute each of them.
def cache_dir():
    """Private to this user: the compiled circuit is a pickle, and a pickle is only loaded from a
    directory nobody else can write to (mode 0700, owned by us); anything else disables the cache."""
    d = os.environ.get("ZKMI_CACHE_DIR") or os.path.join(
        os.environ.get("TMPDIR", "/tmp"), f"zkmi-cache-{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise PermissionError(f"{d}: not owned by this user or writable by others")
    return d


def _source_digest():
    """hash of every Python source of the package: a code change invalidates the cache"""
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "gnark_crypto_primitives_amd")
    for dp, dn, fn in sorted(os.walk(pkg)):
        dn.sort()
        for f in sorted(fn):
            if f.endswith(".py"):
                h.update(f.encode())
                h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()[:16]


def cached(name, make, load, save, use_cache=True):
    """-> (object, hit).  make() under an exclusive lock unless the file is already there."""
    if not use_cache:
        return make(), False
    import fcntl
    try:
        path = os.path.join(cache_dir(), name)
    except (PermissionError, OSError):
        return make(), False
    if os.path.exists(path):
        try:
            return load(path), True
        except Exception:
            pass
    try:
        lk = open(path + ".lock", "w")
    except OSError:
        return make(), False
    with lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if os.path.exists(path):
            try:
                return load(path), True
            except Exception:
                pass
        obj = make()
        tmp = f"{path}.{os.getpid()}.tmp"
        try:                          # a full or read-only cache directory only costs the cache
            save(tmp, obj)
            os.replace(tmp, path)
        except OSError:
            try:
                os.unlink(tmp)
            except OSError:
                pass
        return obj, False


def compile_cached(workload, levels, populated, solver_lanes, use_cache=True):
    """-> (CompiledCircuit, assignment generator, label, cache hit)"""
    import pickle
    from gnark_crypto_primitives_amd import workloads
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    circuit, gen, label = workloads.build(workload, levels, populated)

    def save(path, obj):
        with open(path, "wb") as f:
            pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)

    def load(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    cc, hit = cached(f"cc-{_source_digest()}-{workload}-{levels}-{solver_lanes}.pkl",
                     lambda: compile_circuit(circuit, solver_lanes), load, save, use_cache)
    return cc, gen, label, hit


def _gen_chunk(job):
    workload, levels, populated, seed, count = job
    from gnark_crypto_primitives_amd import workloads
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
    _, gen, _ = workloads.build(workload, levels, populated)
    rng = random.Random(seed)
    return [to_mont_array(_GEN_CC.assignment_vector(gen(rng))) for _ in range(count)]


_GEN_CC = None


def generate_witnesses(cc, workload, levels, jobs, workers, use_cache=True):
    """jobs: [(name, seed, count, populated)] -> {name: [count, n_inputs, 4] uint64 array}; each
    job's array is kept in the start-up cache (keyed by the circuit's fingerprint and the job)."""
    out, fp = {}, cc.fingerprint()
    for job in jobs:
        name, seed, count, populated = job
        out[name], _ = cached(
            f"wit-{fp}-{workload}-{levels}-{populated}-{seed}-{count}.npy",
            lambda: np.stack(_generate(cc, workload, levels, [job], workers)[name])
            if count else np.zeros((0, cc.n_inputs, 4), np.uint64),
            lambda p: np.load(p), lambda p, a: np.save(open(p, "wb"), a), use_cache)
    return out


def _generate(cc, workload, levels, jobs, workers):
    """jobs: [(name, seed, count, populated)] -> {name: [input arrays]}.  Chunks of 32 witnesses
    are dealt to a fork()ed process pool (off-circuit Poseidon in Python big ints is ~0.3 ms per
    hash: 1024 fully populated 160-level paths are 50 s on one core)."""
    global _GEN_CC
    _GEN_CC = cc
    chunks, owner = [], []
    for name, seed, count, populated in jobs:
        for c0 in range(0, count, 32):
            chunks.append((workload, levels, populated, seed * 100003 + c0, min(32, count - c0)))
            owner.append(name)
    out = {name: [] for name, *_ in jobs}
    if workers <= 1 or len(chunks) <= 1:
        res = [_gen_chunk(c) for c in chunks]
    else:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(chunks))) as pool:
            res = pool.map(_gen_chunk, chunks)
    for name, r in zip(owner, res):
        out[name].extend(r)
    return out


def bench_plonk(args):
    """PLONK prove throughput for one workload on one GPU (zkmi_plonk_prove: the five rounds with the
    transcript hashed inside the library); the first proof of the last step is verified with the host
    verifier."""
    from gnark_crypto_primitives_amd import lib, plonk, workloads
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
    from gnark_crypto_primitives_amd.frontend.scs import compile_scs
    t0 = time.time()
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if args.verbose else (lambda *a: None)
    circuit, gen, label = workloads.build(args.workload, args.levels, args.populated)
    sc = compile_scs(compile_circuit(circuit))
    log(f"scs: {sc.n_gates} gates, domain 2^{sc.log_n}, {sc.v_n_steps} solver steps x "
        f"{sc.lanes_per_proof} lanes ({time.time() - t0:.1f}s)")
    B = args.batch
    rng = random.Random(77)
    n_distinct = min(B, args.distinct) if args.distinct > 0 else B
    ws = [to_mont_array(sc.assignment_vector(gen(rng))) for _ in range(n_distinct)]
    inp = np.stack([ws[i % n_distinct] for i in range(B)])
    blind = np.stack([to_mont_array([rng.randrange(workloads.R) for _ in range(9)])
                      for _ in range(B)])
    ctx = lib.Context(0)
    pk = plonk.setup(ctx, sc, 3)
    log(f"setup done ({time.time() - t0:.1f}s)")
    prover = plonk.Prover(ctx, sc, pk, window_bits=args.window_g1, max_batch=max(B, 64))
    log(f"key resident ({time.time() - t0:.1f}s)")
    for _ in range(args.warmup):
        prover.prove_raw(inp, blind)
    ts = time.perf_counter()
    for _ in range(args.steps):
        rec, status = prover.prove_raw(inp, blind)
    elapsed = time.perf_counter() - ts
    proofs = prover.proofs_of(rec[:1])
    n_pub = pk.n_public
    rinv = pow(1 << 256, workloads.R - 2, workloads.R)
    from gnark_crypto_primitives_amd.frontend.compile import array_to_ints
    pub0 = [v * rinv % workloads.R for v in array_to_ints(inp[0, :n_pub])]
    verified = bool(plonk.verify(pk, pub0, proofs[0])) and not status.any()
    out = {"metric": f"proofs/sec, {label}, PLONK/BN254 (KZG)", "value": B * args.steps / elapsed,
           "unit": "proofs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u32x8 Montgomery (254-bit integer)", "data": "synthetic",
           "config": {"workload": f"{label}, batch {B}, PLONK backend", "gates": sc.n_gates,
                      "domain_log2": sc.log_n, "batch_per_gpu": B,
                      "parallelism": "one GPU, blocking rounds, transcript hashed on the host in C++"},
           "first_proof_verifies": verified, "unsatisfied": int((status != 0).sum())}
    print(json.dumps(out))
    prover.close()
    ctx.close()


class stdout_to_stderr:
    """RCCL / gloo print banners on fd 1 when a group comes up; stdout must carry the one JSON
    line only, so fd 1 points at stderr inside this block."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process (which has not touched the GPU
    and never will) warms the start-up cache with every core of the host, then starts N fresh rank
    processes of this same script -- children, not an exec -- and forwards rank 0's stdout."""
    import socket
    import subprocess
    from gnark_crypto_primitives_amd import backend
    n = args.gpus
    t0 = time.time()
    if args.backend == "groth16" and not args.no_cache:
        cc, _, _, _ = compile_cached(args.workload, args.levels, args.populated, args.solver_lanes)
        for r in range(n):
            generate_witnesses(cc, args.workload, args.levels, rank_jobs(args, r, n),
                               args.gen_workers or host_cpus())
        if args.verbose:
            print(f"launcher: start-up cache warm for {n} ranks ({time.time() - t0:.1f}s)",
                  file=sys.stderr, flush=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    sys.exit(backend.launch_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                        n, port))


def rank_jobs(args, rank, world):
    """-> witness jobs [(name, seed, count, populated)] of one rank (seeded per rank)"""
    from gnark_crypto_primitives_amd import backend
    if args.scaling == "strong":
        lo, hi = backend.shard_range(args.batch, rank, world)
        B = hi - lo
    else:
        B = args.batch
    n_distinct = min(B, args.distinct) if args.distinct > 0 else B
    wc_steps = args.worst_case_steps if args.worst_case_steps >= 0 else args.steps
    if args.workload != "arbo" or args.populated >= args.levels - 1:
        wc_steps = 0
    jobs = [("main", 1000 + rank, n_distinct, args.populated)]
    if wc_steps:
        jobs.append(("worst", 5000 + rank, B, args.levels - 1))
    return jobs


def rehearse(args, rank, world, B, global_batch, startup, t0, label):
    """--rehearsal: the N-rank skeleton of a bench run without any GPU work (this is NOT a CPU
    fallback of the prover: no proof is computed, `value` is null).  gloo process group, the same
    barriers, K steps of the per-step all_gather on zeroed 264-byte records."""
    import torch
    import torch.distributed as dist
    from gnark_crypto_primitives_amd import backend
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    with stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    world_seen = dist.get_world_size()
    proofs = torch.zeros((B, 32), dtype=torch.int64)
    proofs[:, 0] = rank                                     # so the gather's placement is checkable
    status = torch.zeros(B, dtype=torch.int32)
    startup["total_s"] = time.time() - t0
    dist.barrier()
    ts = time.perf_counter()
    for _ in range(args.steps):
        gp, gs = backend.gather_proofs(proofs, status, global_batch, force=True)
    dist.barrier()
    elapsed = time.perf_counter() - ts
    lo = backend.shard_range(global_batch, rank, world)[0] if args.scaling == "strong" else rank * B
    ok = bool(gp.shape == (global_batch, 32) and bool((gp[lo:lo + B, 0] == rank).all()))
    t = torch.tensor([int(ok)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({
            "metric": f"proofs/sec, {label}, Groth16/BN254", "value": None, "unit": "proofs/s",
            "rehearsal": True, "n_gpus": world_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "data": "synthetic",
            "config": {"workload": f"{label}, batch {B} proofs per GPU", "global_batch": global_batch,
                       "parallelism": f"rehearsal: gloo process group of {world_seen} rank(s) as "
                                      f"torch.distributed reports it, no GPU work"},
            "gathered_on_every_rank": bool(t.item()), "startup_s": startup}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE if a launcher set it, "
                         "else 1); N > 1 without a launcher starts the N rank processes itself")
    ap.add_argument("--steps", type=int, default=40)
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
                    help="proofs for the CPU baseline (-1: 2 per core, then more up to ~15 s of CPU work; 0: skip)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=-1,
                    help="rehearsal only: put every rank on this device instead of LOCAL_RANK")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch proofs per GPU; strong: --batch proofs in all, sharded")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: still create the (one-rank) process group and run the "
                         "per-step all_gather -- rehearsal of the N > 1 path on a one-GPU box")
    ap.add_argument("--no-gather", action="store_true",
                    help="skip the per-step all_gather of proof records (N > 1)")
    ap.add_argument("--worst-case-steps", type=int, default=-1,
                    help="arbo only: extra timed steps on witnesses with every level populated "
                         "(every wire differs between lanes); -1: as many as --steps, 0: skip")
    ap.add_argument("--gen-workers", type=int, default=0, help="witness generator processes")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one blocking zkmi_prove_batch per step (no overlap of consecutive steps)")
    ap.add_argument("--backend", choices=("groth16", "plonk"), default="groth16",
                    help="plonk: BASELINE config 5's backend (secondary number; one GPU, blocking "
                         "rounds with the Fiat-Shamir hashing on the host between them)")
    ap.add_argument("--msm-chunk-factor", type=int, default=0,
                    help="zkmi_pk_desc.msm_chunk_factor (0 = the library's default, 16 for comb tables)")
    ap.add_argument("--sparse-witness", type=int, default=-1,
                    help="zkmi_pk_desc.sparse_witness (0 dense / 1 mostly small values / 2 all bits); "
                         "-1 = from the circuit's share of boolean wires")
    ap.add_argument("--solver-lanes", type=int, default=0,
                    help="sub-lanes of the witness solver per proof (zkmi_cs_desc.lanes_per_proof); "
                         "0 = the frontend's choice (shortest schedule)")
    ap.add_argument("--entry", choices=("inputs", "witness", "witness-abc"), default="inputs",
                    help="inputs: zkmi_prove_submit on device-resident circuit inputs (the headline). "
                         "witness / witness-abc: the gnark drop-in entry, zkmi_prove_witness_submit "
                         "from HOST memory every step -- solved wire vectors only (R1CS matrices "
                         "resident, a, b, c formed on the device) / wire vectors + a + b + c; the "
                         "timed region then includes the PCIe transfer")
    ap.add_argument("--host-mem", choices=("pageable", "pinned"), default="pageable",
                    help="--entry witness*: where the solved witnesses live (pinned = zkmi_host_alloc)")
    ap.add_argument("--copy-threads", type=int, default=0, help="zkmi_set_copy_threads (0 = default)")
    ap.add_argument("--table-budget-gb", type=float, default=0.0,
                    help="zkmi_pk_desc.table_budget_bytes: cap for the key's MSM tables (0 = "
                         "whatever the free HBM allows); the bounded-memory operating points")
    ap.add_argument("--bounded-gb", type=float, default=64.0,
                    help="after the headline: reload the key with this table budget and time "
                         "--bounded-steps steps (the bounded-memory operating point; 0 = skip)")
    ap.add_argument("--bounded-steps", type=int, default=8)
    ap.add_argument("--no-cache", action="store_true",
                    help="recompute the compiled circuit and the witnesses instead of using the "
                         "start-up cache ($ZKMI_CACHE_DIR, default $TMPDIR/zkmi-cache-<uid>)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="launcher / process-group rehearsal for boxes without a GPU: everything "
                         "but the GPU work (compile, witnesses, rendezvous, barriers, per-step "
                         "all_gather of zeroed records); prints value null and \"rehearsal\": true")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        return launch_ranks(args)
    if env_world is not None and args.gpus is not None and int(env_world) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} disagrees with the inherited WORLD_SIZE={env_world}")
    if args.backend == "plonk":
        if int(env_world or 1) != 1:
            sys.exit("bench.py: --backend plonk is a one-GPU secondary number")
        return bench_plonk(args)

    from gnark_crypto_primitives_amd import backend, workloads
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array

    rank, world, local_rank = backend.env_rank_world()
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if (args.verbose and rank == 0) \
        else (lambda *a: None)
    t0 = time.time()
    use_cache = not args.no_cache
    cc, gen, label, cc_hit = compile_cached(args.workload, args.levels, args.populated,
                                            args.solver_lanes, use_cache)
    startup = {"compile_s": time.time() - t0, "compile_cache_hit": cc_hit}
    log(f"compiled: {cc.n_constraints} constraints, {cc.n_wires} wires, {cc.n_ops} ops in "
        f"{cc.v_n_steps} steps x {cc.lanes_per_proof} lanes "
        f"({time.time() - t0:.1f}s, cache {'hit' if cc_hit else 'miss'})")

    # ---- synthetic witnesses, seeded per rank, generated by worker processes
    # BEFORE this process touches the GPU (nothing is forked once HIP is initialised)
    if args.scaling == "strong":
        g_lo, g_hi = backend.shard_range(args.batch, rank, world)
        B, global_batch = g_hi - g_lo, args.batch
    else:
        B, global_batch = args.batch, args.batch * world
    n_distinct = min(B, args.distinct) if args.distinct > 0 else B
    jobs = rank_jobs(args, rank, world)
    wc_steps = (args.worst_case_steps if args.worst_case_steps >= 0 else args.steps) \
        if len(jobs) > 1 and args.entry == "inputs" else 0
    t1 = time.time()
    sets = generate_witnesses(cc, args.workload, args.levels, jobs, args.gen_workers or
                              max(1, min(16, host_cpus() // world)), use_cache)
    startup["witness_s"] = time.time() - t1
    ws = sets["main"]
    inp_h = np.stack([ws[i % n_distinct] for i in range(B)]) if B else \
        np.zeros((0, cc.n_inputs, 4), np.uint64)
    rng = random.Random(1000 + rank)
    rs_h = np.stack([to_mont_array([rng.randrange(workloads.R), rng.randrange(workloads.R)])
                     for _ in range(max(B, 1))])[:B]
    log(f"witnesses generated ({time.time() - t0:.1f}s)")

    import torch
    import torch.distributed as dist
    if args.rehearsal:
        return rehearse(args, rank, world, B, global_batch, startup, t0, label)
    from gnark_crypto_primitives_amd import groth16, lib

    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # collectives' tensors
    pg = world > 1 or args.force_collective
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        with stdout_to_stderr():
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            # first collective now: RCCL's buffers are allocated before the MSM tables size
            # themselves against the free HBM
            t = torch.zeros(1, dtype=torch.float64, device=cdev)
            dist.all_reduce(t)
            if cdev.type == "cuda":
                torch.cuda.synchronize()

    world_seen = dist.get_world_size() if pg else 1
    if world_seen != world:
        sys.exit(f"bench.py: process group has {world_seen} ranks, WORLD_SIZE says {world}")
    t1 = time.time()
    ctx = lib.Context(local_rank)
    pk, vk, _ = groth16.setup(cc, 2, groth16.gpu_mul(ctx))
    startup["setup_s"] = time.time() - t1
    log(f"setup: log_n={pk.log_n} A={len(pk.a_wire)} B={len(pk.b_wire)} K={len(pk.k_wire)} "
        f"Z={pk.g1_z.shape[0]} ({time.time() - t0:.1f}s)")
    t1 = time.time()
    prover = groth16.Prover(ctx, cc, pk, args.window_g1, args.window_g2, max_batch=max(B, 64),
                            msm_chunk_factor=args.msm_chunk_factor,
                            table_budget_bytes=int(args.table_budget_gb * 1e9),
                            sparse_witness=None if args.sparse_witness < 0 else args.sparse_witness)
    startup["key_load_s"] = time.time() - t1
    log(f"key resident, window tables built ({time.time() - t0:.1f}s)")

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    inp_d, rs_d = to_dev(inp_h), to_dev(rs_h)
    proofs_d = torch.zeros((B, 32), dtype=torch.int64, device=dev)
    status_d = torch.zeros(B, dtype=torch.int32, device=dev)
    # keys with the commitment extension also return proof.Commitments and proof.CommitmentPok
    coms_d = torch.zeros((B, prover.n_commitments + 1, 8), dtype=torch.int64, device=dev) \
        if prover.n_commitments else None
    torch.cuda.synchronize()
    log(f"witnesses resident ({time.time() - t0:.1f}s)")
    gather = pg and not args.no_gather
    gathered = [None]
    wit = None
    if args.entry != "inputs" and B:
        # solved witnesses in host memory, as gnark's solver would leave them (untimed set-up: the
        # GPU solver's output copied out once)
        t1 = time.time()
        status, W, abc = prover.solve(inp_h, want_wires=True, want_abc=args.entry == "witness-abc")
        assert not status.any()
        arrs = [W] + ([abc[0], abc[1], abc[2]] if abc is not None else [])
        if args.host_mem == "pinned":
            pinned = []
            for x in arrs:
                y = ctx.host_alloc(x.shape)
                y[...] = x
                pinned.append(y)
            arrs = pinned
        if args.copy_threads:
            ctx.set_copy_threads(args.copy_threads)
        if args.entry == "witness":
            prover.load_r1cs()
        wit = arrs
        startup["witness_solve_s"] = time.time() - t1
        log(f"solved witnesses on the host: {sum(x.nbytes for x in arrs) / 1e9:.2f} GB "
            f"({args.host_mem}) ({time.time() - t0:.1f}s)")

    def prove_blocking(inp, rsd):
        if coms_d is None:
            prover.prove(inp, rsd, proofs_d, status_d)
        else:
            prover.submit(inp, rsd)
            prover.collect(proofs_d, status_d, coms_d)

    def submit(inp, rsd):
        if wit is None:
            prover.submit(inp, rsd)
        else:
            prover.submit_witness(wit[0], rs_h, *wit[1:])

    def finish_step():
        """the path's only exchange: all_gather of this step's proof records (RCCL on device
        tensors; gloo rehearsals stage through the host)"""
        if not gather:
            return
        if cdev.type == "cuda":
            gathered[0] = backend.gather_proofs(proofs_d, status_d, global_batch, force=True)
        else:
            gathered[0] = backend.gather_proofs(proofs_d.cpu().numpy().view(np.uint64),
                                                status_d.cpu().numpy(), global_batch, "cpu",
                                                force=True)

    def timed_steps(steps, inp, rsd):
        """K steps bracketed by barrier + synchronize on both sides; returns (max-over-ranks
        seconds, per-stage ms summed over the steps)."""
        stage = np.zeros(8)
        if pg:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        if B == 0:
            for _ in range(steps):
                finish_step()
        elif args.no_pipeline:
            for _ in range(steps):
                if wit is None:
                    prove_blocking(inp, rsd)
                else:
                    submit(inp, rsd)
                    prover.collect(proofs_d, status_d, coms_d)
                stage += np.array(ctx.last_timings())
                finish_step()
        else:
            submit(inp, rsd)
            for k in range(steps):
                if k + 1 < steps:
                    submit(inp, rsd)
                prover.collect(proofs_d, status_d, coms_d)
                stage += np.array(ctx.last_timings())
                finish_step()
        torch.cuda.synchronize()
        if pg:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        if pg:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, stage

    startup["total_s"] = time.time() - t0
    # Steps are software-pipelined two deep through the library's submit/collect pair: the
    # latency-bound witness solve of step k+1 (16 wavefronts) runs on a second HIP stream under the
    # NTT/MSM kernels of step k.  The timed region contains K submits and K collects: the first
    # solve is exposed, nothing of the timed work happens outside the region.
    for _ in range(args.warmup):
        if B and wit is None:
            prove_blocking(inp_d, rs_d)
        elif B:
            submit(inp_d, rs_d)
            prover.collect(proofs_d, status_d, coms_d)
        finish_step()
    elapsed, stage = timed_steps(args.steps, inp_d, rs_d)
    status = status_d.cpu().numpy()
    n_bad = int((status != 0).sum())
    if pg:
        t = torch.tensor([n_bad], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        n_bad = int(t.item())
    proofs = proofs_d.cpu().numpy().view(np.uint64)
    gather_ok = None
    if gather:
        gp, gs = gathered[0]
        gp = gp.cpu().numpy().view(np.uint64) if hasattr(gp, "cpu") else gp
        lo = backend.shard_range(global_batch, rank, world)[0] if args.scaling == "strong" \
            else rank * B
        gather_ok = bool(gp.shape == (global_batch, 32) and np.array_equal(gp[lo:lo + B], proofs)
                         and int((np.asarray(gs.cpu() if hasattr(gs, "cpu") else gs) != 0).sum())
                         == n_bad)
    # worst case for the table gathers: every level populated, so every wire differs between lanes
    worst = None
    if wc_steps:
        inp_w = to_dev(np.stack(sets["worst"]))
        prove_blocking(inp_w, rs_d)
        e_w, st_w = timed_steps(wc_steps, inp_w, rs_d)
        bad_w = int((status_d.cpu().numpy() != 0).sum())
        worst = {"value": global_batch * wc_steps / e_w, "steps": wc_steps,
                 "populated": args.levels - 1, "ms_per_step": e_w / wc_steps * 1e3,
                 "msm_g1_kernel_only_ms": st_w[6] / wc_steps, "unsatisfied": bad_w}
    # bounded-memory operating point: the same key with its MSM tables capped (a GPU shared with
    # other circuits); N = 1, headline workload and entry only
    bounded = None
    main_info = ctx.pk_info(prover.pk_h)
    if args.bounded_gb > 0 and world == 1 and B and wit is None and not args.no_pipeline and \
            args.workload == "arbo" and \
            args.table_budget_gb == 0 and args.window_g1 == 0 and args.window_g2 == 0:
        prover.close()
        prover = groth16.Prover(ctx, cc, pk, 0, 0, max_batch=max(B, 64),
                                table_budget_bytes=int(args.bounded_gb * 1e9))
        info_b = ctx.pk_info(prover.pk_h)
        prove_blocking(inp_d, rs_d)
        e_b, _ = timed_steps(args.bounded_steps, inp_d, rs_d)
        same = bool(np.array_equal(proofs_d.cpu().numpy().view(np.uint64), proofs))
        bounded = {"table_budget_gb": args.bounded_gb, "value": B * args.bounded_steps / e_b,
                   "steps": args.bounded_steps,
                   "table_bytes": info_b["g1_table_bytes"] + info_b["g2_table_bytes"],
                   "g1_comb_k": info_b["g1_comb_k"], "g2_comb_k": info_b["g2_comb_k"],
                   "proofs_equal_unbounded_run": same}
    if rank == 0:
        stage /= max(args.steps, 1)
        # ---- roofline of the dominant kernel: the G1 accumulate kernel, four launches per step
        # (msm_accumulate_comb<Fq, false> under the default comb plan; stage[6] = sum of the HIP
        # event pairs that bracket each accumulate launch alone)
        ns = [len(pk.a_wire), len(pk.b_wire), len(pk.k_wire), pk.g1_z.shape[0]]
        alg_bytes = sum(n * 64 + B * n * 32 for n in ns)
        msm_s = stage[6] * 1e-3
        achieved = alg_bytes / msm_s / 1e9 if msm_s > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE +
        # WRITE_SIZE of the four launches, raw counter values; same circuit, batch and windows)
        info = main_info
        traffic, traffic_source = None, None
        try:
            import glob
            pm_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))[-1]
            pm = json.load(open(pm_path))
            if pm.get("batch") == B and pm.get("levels") == args.levels and \
                    args.workload == "arbo" and pm.get("g1_comb_k") == info["g1_comb_k"] and \
                    pm.get("populated") == args.populated:
                traffic = pm["msm_g1_bytes_per_launch"]
                traffic_source = (f"profiles/{os.path.basename(pm_path)} (rocprofv3 --pmc FETCH_SIZE / "
                                  "WRITE_SIZE passes of tools/pmc_collect.sh over this build and "
                                  "plan; not collected inside this run)")
        except (OSError, ValueError, KeyError, IndexError):
            pass
        kname = ("msm_accumulate_comb<Fq, false>" if info["g1_comb_k"] else
                 "msm_accumulate_shared<Fq>" if info["g1_shared"] else "msm_accumulate<Fq, false>")
        roofline = {"bound": "hbm", "kernel": f"{kname} (G1 MSMs of the key)",
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
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
        if (info["g1_comb_k"] and info["g1_windows"] == 254) or cc.n_boolean_wires * 2 > cc.n_wires:
            # subset-sum tables skip every all-zero digit (bit-valued witnesses skip most of them):
            # the additions actually executed are not counted, so this view is only an upper bound
            roofline["alu"]["frac"] = None
            roofline["alu"]["note"] = ("subset-sum tables: zero digits are skipped, `achieved` "
                                       "counts every (group, window) and is an upper bound")
        try:
            roofline["alu"]["fq_mul_per_s_ff29_microbench"] = ctx.field_mul_bench(2, 1 << 22, 256)
        except Exception:
            pass
        cpu = None
        if args.cpu_sample != 0 and world == 1:      # CPU baseline: rank 0 at N = 1 only
            from oracle import cref
            cores = available_cpus()
            rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
            ckh = cref.CommitKeysHandle(pk) if prover.n_commitments else None

            def cpu_prove(lo, hi):
                t = time.perf_counter()
                if ckh is not None:
                    want, wcoms, wpoks, wstatus, used = cref.groth16_prove_batch_ex(
                        rh, ph, ckh, inp_h[lo:hi], rs_h[lo:hi], cores)
                    got_c = coms_d.cpu().numpy().view(np.uint64)[lo:hi]
                    ok = bool(np.array_equal(got_c[:, :-1], wcoms) and np.array_equal(got_c[:, -1], wpoks))
                else:
                    want, wstatus, used = cref.groth16_prove_batch(rh, ph, inp_h[lo:hi], rs_h[lo:hi], cores)
                    ok = True
                t = time.perf_counter() - t
                return t, used, bool(ok and np.array_equal(want, proofs[lo:hi]) and not wstatus.any())

            # default sample: 2 proofs per core, then as many more as bring the CPU work to ~15 s
            S = args.cpu_sample if args.cpu_sample > 0 else min(B, 2 * cores)
            tc, used, same = cpu_prove(0, S)
            if args.cpu_sample < 0 and tc < 10.0 and S < B:
                more = min(B - S, int(S * (15.0 - tc) / tc) // cores * cores)
                if more > 0:
                    t2, used, same2 = cpu_prove(S, S + more)
                    tc, S, same = tc + t2, S + more, same and same2
            cpu = {"value": S / tc, "unit": "proofs/s", "cores": used, "kind": "port",
                   "sample": f"first {S} proofs of the same batch, C oracle (oracle/c), "
                             f"OpenMP over proofs", "seconds": tc,
                   "gpu_proofs_bit_exact_vs_cpu": same}
        out = {
            "metric": "proofs/sec, Arbo-160 Poseidon SMT-verifier circuit, Groth16/BN254"
                      if args.workload == "arbo" and args.levels == 160
                      else f"proofs/sec, {label}, Groth16/BN254",
            "value": global_batch * args.steps / elapsed, "unit": "proofs/s",
            "n_gpus": world_seen,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u32x8 Montgomery (254-bit integer)",
            "data": "synthetic",
            "config": {"workload": f"{label}, batch {B} proofs per GPU"
                                   + (f" (global batch {global_batch} sharded)"
                                      if args.scaling == "strong" else ""),
                       "constraints": cc.n_constraints, "wires": cc.n_wires,
                       "domain_log2": pk.log_n, "batch_per_gpu": B,
                       "msm_terms_per_proof": {"g1": int(sum(ns)), "g2": ns[1]},
                       "msm_window_tables": main_info,
                       "global_batch": global_batch,
                       "parallelism": (f"batch-split x{world} ({args.scaling} scaling), "
                                       f"{args.dist_backend if pg else 'no'} process group "
                                       f"of {world_seen} rank(s) as torch.distributed reports it"
                                       + (", all_gather of 264-B proof records per step"
                                          if gather else ", no collective in the data path"))},
            "gathered_on_every_rank": gather_ok,
            "value_worst_case": worst["value"] if worst else None,
            "worst_case": worst,
            "bounded": bounded,
            "pipelined": not args.no_pipeline,
            "entry": {"inputs": "zkmi_prove_submit (circuit inputs resident in HBM)",
                      "witness": "zkmi_prove_witness_submit, wire vectors from host memory, "
                                 "a, b, c formed on the device (zkmi_r1cs_load)",
                      "witness-abc": "zkmi_prove_witness_submit, W + a + b + c from host memory"
                      }[args.entry],
            "host_transfer": None if wit is None else {
                "memory": args.host_mem, "bytes_per_step": int(sum(x.nbytes for x in wit)),
                "GB_per_s_if_serial": sum(x.nbytes for x in wit) / 1e9 / (elapsed / args.steps)},
            "stage_ms": {"solve": stage[0], "quotient_ntt": stage[1], "msm_g1": stage[2],
                         "msm_g2": stage[3], "assemble_overlapped": stage[4], "main_stream_span": stage[5],
                         "msm_g1_kernel_only": stage[6], "msm_g2_kernel_only": stage[7]},
            "unsatisfied": n_bad,
            "startup_s": startup,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    prover.close()
    ctx.close()
    if pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
