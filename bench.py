"""bench.py -- BASELINE.json metric on MI355X: decode tokens/s (+ prefill TFLOP/s) of Llama-3-8B fp8 (w8a8),
batch 32 x seq 2048, synthetic random weights and token ids, every device op a hand-written HIP kernel.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: this process starts the N ranks itself, one per GPU, as bench_one_batch.py:527-546 spawns one
     process per tp_rank, and relays rank 0's JSON line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py
     --gpus N ...` the launcher's RANK / LOCAL_RANK / WORLD_SIZE are used and must agree with --gpus)

A "step" is one decode step of the whole model for the whole batch (32 new tokens).  Inputs (weights, the KV pool
filled by a real prefill of 32 x 2048 tokens, index tensors) are resident in HBM before the timed region.
Prints ONE JSON line (rank 0).  N > 1 shards the model tensor-parallel exactly like the reference
(linear.py / llama.py:118-133) with one RCCL all-reduce per row-parallel layer: strong scaling of the same batch.

Extra objects on the line:
  roofline     dominant kernel = decode_attn_stage1 (KV stream).  achieved = algorithmic KV bytes of one launch
               (bs * seq * Hkv * D * 2 B * 2) / mean launch duration measured here with HIP events on the launch stream.
  cpu_baseline the oracle's torch-native restatement timed on this box's host cores on a bounded sample
               (one of the 32 layers at the full batch 32 x 2048, extrapolated to the step) -- a baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b", "qwen2-7b", "tiny"])
    ap.add_argument("--quant", default="w8a8_fp8", choices=["w8a8_fp8", "fp8", "awq", "none"])
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; invalidates the metric)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="activation / KV dtype (AWQ checkpoints usually run f16)")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e4m3"],
                    help="auto = the model dtype (the BASELINE configuration); fp8_e4m3 halves the KV stream (reported separately)")
    ap.add_argument("--no-graph-metadata", action="store_true", help="measurement hook: keep the decode metadata launches outside the captured graph")
    ap.add_argument("--decode-attn-mode", type=int, default=-1, help="measurement hook: 0 / 1 = sgl_mi355_decode_attention_set_mode")
    ap.add_argument("--no-pad-qkv", action="store_true", help="measurement hook: the prefill's qkv output rows at their natural stride")
    ap.add_argument("--extend-attn-mode", type=int, default=-1, help="measurement hook: sgl_mi355_extend_attention_set_mode (0 / 2 / 3 = the older kernels, 5 = force the 32x32x16 kernel)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo = rehearsal of the N > 1 path on fewer GPUs")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--allow-eager", action="store_true",
                    help="if the HIP-graph capture of the decode step fails, time eager (launch-bound) steps and say so in the metric "
                         "name instead of exiting with code 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused-decode", action="store_true", help="measurement hook: the per-op decode path (same logits, more launches)")
    ap.add_argument("--prefill-chunk", type=int, default=32,
                    help="requests per prefill call (32 = the whole BASELINE batch in one extend, as bench_one_batch.py does; "
                         "sweep on one box: 4 / 8 / 16 / 32 requests -> 1 915 / 1 950 / 1 958 / 1 978 TFLOP/s)")
    ap.add_argument("--gemm-hook", type=int, default=0, help="measurement hook: value passed to sgl_mi355_fp8_gemm_force_tile")
    ap.add_argument("--skinny-hook", type=int, default=0, help="measurement hook: value passed to sgl_mi355_skinny_gemm_force_generic "
                    "(2 = 8-row tiles one tile ahead, the round-3 form; 3 = all tiles of a workgroup up front, the default; 4 / 5 = slab-mode "
                    "launches with one / two tiles in flight per wave)")
    ap.add_argument("--kv-split-rule", type=int, default=3, help="0 = the reference's heuristic, 1 = max splits everywhere, 2 = the MI355X balance rule, 3 = rule 2's units as a list sorted longest first, on a grid without never-live workgroups")
    ap.add_argument("--kv-sched-rounds-pct", type=int, default=150, help="rule 3: units sized for this many percent of one round of resident workgroups")
    ap.add_argument("--max-kv-splits", type=int, default=16, help="triton_attention_num_kv_splits (16 = the reference's HIP default)")
    ap.add_argument("--all-reduce", default="auto", choices=["auto", "rccl", "p2p"],
                    help="N > 1: auto = the one-shot P2P all-reduce over IPC-mapped peer buffers when its start-up self-check against "
                         "RCCL passes, else RCCL; rccl / p2p force one")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json configs[N - 1] as the survey defines it (one driver-runnable command per configuration, "
                         "bench_one_batch.py:527-546): 1 = the CPU plumbing case (oracle, no GPU); 2 = Llama-3-8B bf16 32 x 2048; "
                         "3 = Llama-3-8B w8a8 batch 64, 1536 shared + 512 private tokens through the radix hit path and cascade decode; "
                         "4 = Qwen2-7B int4 AWQ f16 32 x 1024; 5 = rank 0's shard of Llama-3-70B fp8 TP 8 (one process, collectives = "
                         "same-size copies), batch 128, ragged lengths U(512, 4096).  0 = the flags as given (default: the headline).")
    ap.add_argument("--seq-dist", default="uniform", choices=["uniform", "ragged"],
                    help="ragged: per-request lengths drawn uniformly from [--seq-min, --seq-len] with a fixed seed, as "
                         "bench_serving.py:1013 sample_random_requests draws them for every throughput run")
    ap.add_argument("--seq-min", type=int, default=0, help="ragged: shortest request (default: --seq-len / 4)")
    ap.add_argument("--shared-prefix", type=int, default=0,
                    help="P > 0: every request = one common prefix of P tokens + (--seq-len - P) private tokens; the prefill goes through "
                         "the radix cache (request 0 in full, the others extend over the cached prefix), decode uses cascade attention")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="--gpus N without a launcher: wall-clock limit of the whole job (s)")
    ap.add_argument("--emulate-tp", type=int, default=0,
                    help="ONE process builds rank 0's shard of a tp-N model and replaces each collective by a same-size device copy: "
                         "per-rank compute ms/step without communication (TP-readiness measurement on a 1-GPU box; not the metric)")
    args = ap.parse_args()
    preset = {2: dict(model="llama3-8b", quant="none", batch=32, seq_len=2048),
              3: dict(model="llama3-8b", quant="w8a8_fp8", batch=64, seq_len=2048, shared_prefix=1536),
              4: dict(model="qwen2-7b", quant="awq", batch=32, seq_len=1024, dtype="f16"),
              5: dict(model="llama3-70b", quant="w8a8_fp8", batch=128, seq_len=4096, seq_min=512, seq_dist="ragged", emulate_tp=8)}.get(args.config)
    for k, v in (preset or {}).items():
        setattr(args, k, v)
    if args.seq_dist == "ragged" and args.seq_min <= 0:
        args.seq_min = max(1, args.seq_len // 4)
    if args.shared_prefix and not (0 < args.shared_prefix < args.seq_len and args.seq_dist == "uniform"):
        ap.error("--shared-prefix P needs 0 < P < --seq-len and uniform lengths")
    return args


def config1_cpu_plumbing(args):
    """BASELINE configs[0]: greedy decode, batch 1, torch-native attention backend on the CPU -- the reference's own CPU-runnable case
    (plumbing, no GPU).  Here: the oracle's restatement of that stack (oracle/model.py, torch-native attention) on a small Llama-shaped
    network, a few greedy steps on the host cores; the parity of this path against the reference's goldens is tests/test_oracle_model.py."""
    from oracle.model import OracleLlama
    from types import SimpleNamespace

    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    cfg = SimpleNamespace(hidden_size=768, num_attention_heads=12, num_key_value_heads=12, head_dim=64, num_hidden_layers=12,
                          intermediate_size=3072, vocab_size=50272, max_position_embeddings=2048, rope_theta=10000.0, rms_norm_eps=1e-5)
    g = torch.Generator().manual_seed(0)

    def w(n, k):
        return ((torch.randn(n, k, generator=g) * 0.02).bfloat16(), None)

    hq, d, hid, inter = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size
    weights = {"embed": (torch.randn(cfg.vocab_size, hid, generator=g) * 0.02).bfloat16(), "lm_head": (torch.randn(cfg.vocab_size, hid, generator=g) * 0.02).bfloat16(),
               "norm": torch.ones(hid).bfloat16(),
               "layers": [{"ln1": torch.ones(hid).bfloat16(), "ln2": torch.ones(hid).bfloat16(),
                           "qkv": w(3 * hq * d, hid) + (None,), "o": w(hid, hq * d), "gate_up": w(2 * inter, hid), "down": w(hid, inter)}
                          for _ in range(cfg.num_hidden_layers)]}
    prompt, steps = 128, max(1, min(args.steps, 16))
    m = OracleLlama(cfg, weights, torch.bfloat16, False, prompt + steps + 2)
    r2t = torch.arange(1, prompt + steps + 2, dtype=torch.int32).view(1, -1)
    rpi, ids = torch.zeros(1, dtype=torch.int64), torch.randint(0, 10000, (prompt,), generator=g)
    seq = torch.tensor([prompt], dtype=torch.int64)
    logits = m.forward(ids, torch.arange(prompt), r2t, rpi, seq, r2t[0, :prompt].long(), torch.zeros(1, dtype=torch.int32), torch.tensor([prompt], dtype=torch.int32))
    nxt = logits[-1:].float().argmax(-1)
    t0 = time.perf_counter()
    for i in range(steps):
        seq = seq + 1
        logits = m.forward(nxt, seq - 1, r2t, rpi, seq, r2t[0, seq[0] - 1].view(1).long())
        nxt = logits[-1:].float().argmax(-1)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"metric": "decode tokens/sec, OPT-125m-sized network, greedy, batch 1, torch-native attention on the CPU (BASELINE configs[0]: plumbing, no GPU)",
                      "value": 1.0 / dt, "unit": "tokens/s", "n_gpus": 0, "steps": steps, "warmup": 0, "ms_per_step": dt * 1e3, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": "OPT-125m-sized (12 layers, hidden 768, 12 heads x 64) greedy decode, batch 1, prompt 128, oracle/model.py on the host",
                                 "baseline_config": 1, "global_batch": 1, "seq_len": prompt, "parallelism": "cpu"},
                      "roofline": None, "cpu_baseline": {"value": 1.0 / dt, "unit": "tokens/s", "cores": cores, "kind": "port", "sample": "the measurement itself"}}))


def cpu_baseline_sample(cfg, batch, seq_len, layers):
    """One decoder layer's hot path at the full batch on the host cores, with the oracle's restatements:
    torch-native decode attention (per-request gather + SDPA) + the four w8a8 fp8 linears (reference test formula)."""
    from oracle import attention as oa
    from oracle import quant as oq

    # the box's CPU share, not the host's core count (a 1-GPU box is given 16 cores' worth)
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    hq, hkv, d, hid, inter = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size
    slots = batch * seq_len + 1
    k_buf = torch.randn(slots, hkv, d, generator=g).bfloat16()
    v_buf = torch.randn(slots, hkv, d, generator=g).bfloat16()
    req_to_token = (torch.randperm(slots - 1, generator=g) + 1).int().view(batch, seq_len)
    rpi = torch.arange(batch)
    seq = torch.full((batch,), seq_len, dtype=torch.int64)
    q = torch.randn(batch, hq, d, generator=g).bfloat16()
    shapes = [(hq * d + 2 * hkv * d, hid), (hid, hq * d), (2 * inter, hid), (hid, inter)]
    ws = []
    for n, k in shapes:
        w = (torch.randn(n, k, generator=g) * 0.02)
        s = w.abs().amax(1, keepdim=True) / 448.0
        ws.append(((w / s).to(torch.float8_e4m3fn), s.flatten()))
    xs = [torch.randn(batch, k, generator=g).bfloat16() for _, k in shapes]

    def layer_once():
        oa.decode_attention_sdpa(q, k_buf, v_buf, req_to_token, rpi, seq, d ** -0.5)
        for (wq, sw), x in zip(ws, xs):
            xq, sx = oq.per_token_quant_fp8(x)
            oq.scaled_mm(xq, wq.t(), sx.flatten(), sw, torch.bfloat16)

    def median_of(fn, n=5):
        fn()  # warm-up
        times = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return sorted(times)[len(times) // 2], len(times)

    t_layer, n_dec = median_of(layer_once)
    step = t_layer * layers
    # extend leg: one layer's torch-native extend attention + the four linears at a reduced batch
    eb = min(2, batch)   # reduced batch: the full 32 x 2048 extend would take minutes per iteration on the host
    e_seq = torch.full((eb,), seq_len, dtype=torch.int64)
    e_q = torch.randn(eb * seq_len, hq, d, generator=g).bfloat16()
    e_xs = [torch.randn(eb * seq_len, k, generator=g).bfloat16() for _, k in shapes]
    zeros = torch.zeros(eb, dtype=torch.int32)

    def extend_once():
        oa.extend_attention_sdpa(e_q, k_buf, v_buf, req_to_token, rpi[:eb], e_seq, zeros, e_seq.int(), d ** -0.5)
        for (wq, sw), x in zip(ws, e_xs):
            xq, sx = oq.per_token_quant_fp8(x)
            oq.scaled_mm(xq, wq.t(), sx.flatten(), sw, torch.bfloat16)

    t_ext, n_ext = median_of(extend_once)
    lin = sum(n * k for n, k in shapes)
    ext_flops = 2.0 * lin * eb * seq_len + eb * (4.0 * seq_len * seq_len * hq * d) / 2
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": batch / step, "unit": "tokens/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"1 of {layers} decoder layers (torch-native decode attention + 4 w8a8 linears) at batch {batch} x seq {seq_len}, "
                      f"median of {n_dec} = {t_layer:.3f} s, extrapolated x{layers}; lm_head/norms excluded",
            "extend": {"value": ext_flops / t_ext / 1e12, "unit": "TFLOP/s",
                       "sample": f"1 layer (torch-native extend attention + 4 w8a8 linears) at batch {eb} x seq {seq_len} without prefix, "
                                 f"median of {n_ext} = {t_ext:.3f} s"}}


def _traffic_profile():
    """Newest committed PMC summary (profiles/round*_pmc_traffic.json) and the git SHA it was taken at."""
    import glob
    import re

    files = glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_traffic.json"))
    if not files:
        return None, None
    # (round number, then name: a re-profile later in a round is "round3b_..." and sorts after "round3_...")
    path = max(files, key=lambda f: (int(re.search(r"round(\d+)", os.path.basename(f)).group(1)), os.path.basename(f)))
    with open(path) as f:
        return os.path.relpath(path, ROOT), json.load(f)


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here -- one child process per GPU with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set, the reference's bench_one_batch.py:527-546 (one process per tp_rank) -- relay rank 0's
    stdout (the JSON line), and exit non-zero as soon as any rank does.  This parent never touches a GPU (no HIP call, no torch.cuda
    call, no exec of a process that has initialised one): the children are fresh interpreters in their own sessions, and every exit
    path of the parent -- a failed rank, SIGTERM / SIGINT from a driver's timeout, the wall-clock limit, an exception -- ends them."""
    import glob
    import signal
    import socket
    import subprocess

    n = args.gpus
    # A profiler's preloaded library has initialised the GPU in THIS process already: starting N children from it is the exec hop the
    # GPU pool forbids (and the counters would be per process anyway).  Profile one rank: `rocprofv3 ... -- python3 bench.py --gpus 1`.
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise SystemExit("[bench] refusing to start the ranks of --gpus N under a profiler (its preloaded library has initialised the GPU "
                         "in this process); profile a --gpus 1 run, or launch the ranks with torch.distributed.run")
    if args.dist_backend == "nccl":
        # GPUs from the KFD topology (no HIP / torch.cuda call in the parent): nodes with a non-zero simd_count
        gpus = 0
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            try:
                with open(f) as fh:
                    gpus += any(ln.split()[0] == "simd_count" and int(ln.split()[1]) > 0 for ln in fh if ln.strip())
            except OSError:
                pass
        if 0 < gpus < n:   # (0: topology not readable here -- let the ranks fail fast themselves)
            raise SystemExit(f"[bench] --gpus {n} over RCCL needs {n} visible GPUs, the KFD topology lists {gpus} "
                             f"(--dist-backend gloo rehearses the N > 1 path on fewer)")
    with socket.socket() as sk:   # (bind-then-close can lose the port to another process in between: the ranks then fail at rendezvous
        sk.bind(("127.0.0.1", 0))   #  and the job exits non-zero -- rerun; a launcher-provided MASTER_PORT avoids it)
        port = sk.getsockname()[1]
    procs = []

    def end_children(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)   # the child's own session: itself and anything it started
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, frame):
        end_children()
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL and the P2P all-reduce both need it on this driver
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True,
                                          stdout=None if r == 0 else sys.stderr))   # only rank 0 owns stdout
        deadline = time.monotonic() + args.spawn_timeout
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    if code != 0:
                        rc = code if code > 0 else 1
            if time.monotonic() > deadline:
                print(f"[bench] the ranks did not finish within --spawn-timeout {args.spawn_timeout:.0f} s", file=sys.stderr)
                rc = 124
    finally:
        end_children()   # a rank failed, the limit passed or this process is leaving: peers would wait in a collective forever
        t_end = time.monotonic() + 20
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                pass
        end_children(signal.SIGKILL)
        for sg, h in old.items():
            signal.signal(sg, h)
    if rc:
        print(f"[bench] a rank exited with code {rc}; the job is void", file=sys.stderr)
    raise SystemExit(rc)


def main():
    args = parse_args()
    if args.config == 1:
        return config1_cpu_plumbing(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)   # does not return
    phases, t_phase = {}, time.perf_counter()

    def phase(name):
        nonlocal t_phase
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - t_phase, 3)
        t_phase = now

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: the two must agree "
                         f"(plain `python bench.py --gpus N` starts its own N ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    local_rank %= max(1, torch.cuda.device_count())  # (gloo rehearsal: several ranks may share one GPU)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"[bench] process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    from __graft_entry__ import load_package

    load_package()
    from ltp_sglang_amd.srt.distributed import communication_op as comm
    from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner

    comm.init_tensor_parallel()
    cfg = {"llama3-8b": LlamaShape.llama3_8b, "llama3-70b": LlamaShape.llama3_70b, "qwen2-7b": LlamaShape.qwen2_7b,
           "tiny": LlamaShape.tiny}[args.model]()
    if args.layers:
        cfg.num_hidden_layers = args.layers
    ar_kind = "none" if world == 1 else ("rccl" if args.dist_backend == "nccl" else "gloo through host copies (rehearsal)")
    if world > 1 and args.all_reduce != "rccl" and (args.dist_backend == "nccl" or args.all_reduce == "p2p"):
        # one-shot P2P all-reduce / all-gather (custom_all_reduce_hip.cuh:261-294 in the reference): IPC handles travel over a
        # gloo side group; before it is trusted, one all-reduce and one all-gather are compared with RCCL's results on every rank
        import torch.distributed as dist

        from ltp_sglang_amd.srt.distributed.custom_all_reduce import CustomAllreduce

        ok, why = True, ""
        try:
            # (gloo rehearsal on one GPU: the WORLD group is already a CPU group)
            car = CustomAllreduce(dist.new_group(backend="gloo") if args.dist_backend == "nccl" else dist.group.WORLD, torch.device(dev))
            ok = not car.disabled
            why = car.disabled_reason
            if ok:
                probe = torch.randn(32, 4096, device=dev, generator=torch.Generator(device=dev).manual_seed(rank)).to(torch.bfloat16)
                want = probe.float()
                want = comm.tensor_model_parallel_all_reduce(want)   # f32 sum over RCCL (gloo rehearsal: host-staged); the P2P kernel also accumulates in f32
                got = car.all_reduce(probe.clone())
                car.check_error()
                ok = bool(((got.float() - want).abs() <= 0.02 * want.abs() + 0.05).all())
                shard = torch.full((4, 64), float(rank), device=dev, dtype=torch.bfloat16)
                gathered = car.all_gather_last_dim(shard)
                car.check_error()
                ok = ok and all(bool((gathered[:, r * 64:(r + 1) * 64] == float(r)).all()) for r in range(world))
                if ok:   # the form the decode step uses: all-reduce + residual add + RMSNorm + per-token fp8 quant in one launch
                    from ltp_sglang_amd.sgl_kernel import fused_add_rmsnorm_quant_fp8

                    gw = (1 + 0.1 * torch.randn(4096, device=dev, generator=torch.Generator(device=dev).manual_seed(7))).to(torch.bfloat16)
                    res0 = torch.randn(32, 4096, device=dev, generator=torch.Generator(device=dev).manual_seed(11)).to(torch.bfloat16)
                    r1, r2 = res0.clone(), res0.clone()
                    n1, _, s1 = fused_add_rmsnorm_quant_fp8(want.to(torch.bfloat16), r1, gw, 1e-5, want_norm=True)
                    n2, _, s2 = car.all_reduce_add_rmsnorm_quant(probe.clone(), r2, gw, 1e-5, want_norm=True)
                    car.check_error()
                    ok = (bool(((n2.float() - n1.float()).abs() <= 0.03 * n1.float().abs() + 0.06).all())
                          and bool(((r2.float() - r1.float()).abs() <= 0.03 * r1.float().abs() + 0.06).all())
                          and bool(((s2 - s1).abs() <= 0.03 * s1.abs() + 1e-6).all()))
                why = "" if ok else "self-check against RCCL failed"
        except Exception as e:   # IPC not available between these devices, driver limits, ...
            ok, why = False, f"{type(e).__name__}: {e}"
        flag = torch.tensor([1 if ok else 0], device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # all ranks or none
        if int(flag.item()) == 1:
            # The two-stage kernels (custom_all_reduce_hip.cuh:294) become part of the default dispatch only after THIS job has
            # checked them, plain and fused, on a message above the reference's threshold (>= 1 MiB at the model's hidden size)
            # against RCCL on every rank; otherwise every message stays on the one-shot kernel.
            two_stage = car.validate_two_stage(comm.tensor_model_parallel_all_reduce, rows=max(128, (1 << 19) // cfg.hidden_size),
                                               hidden=cfg.hidden_size)
            comm.set_custom_all_reduce(car)
            ar_kind = ("p2p (one-shot below the reference's size rule, two-stage above -- both self-checked against "
                       + ("RCCL" if args.dist_backend == "nccl" else "the host sum") + "; fused with add + RMSNorm + quant)"
                       if two_stage else "p2p one-shot at every size (the two-stage self-check did not pass); fused with add + RMSNorm + quant")
        elif args.all_reduce == "p2p":
            raise SystemExit(f"[bench] rank {rank}: --all-reduce p2p requested but unusable ({why or 'a peer failed'})")
        elif rank == 0:
            print(f"[bench] one-shot P2P all-reduce not used ({why or 'a peer failed'}); RCCL instead", file=sys.stderr)
    if args.emulate_tp > 1:
        if world > 1:
            raise SystemExit("--emulate-tp is a single-process measurement")
        comm.init_emulated_tensor_parallel(args.emulate_tp)
    if args.gemm_hook:
        from ltp_sglang_amd import _cabi

        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(args.gemm_hook))
    if args.skinny_hook:
        from ltp_sglang_amd import _cabi

        _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(args.skinny_hook))
    if args.decode_attn_mode >= 0:
        from ltp_sglang_amd import _cabi

        _cabi.check(_cabi.lib.sgl_mi355_decode_attention_set_mode(args.decode_attn_mode))
    if args.extend_attn_mode >= 0:
        from ltp_sglang_amd import _cabi

        _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(args.extend_attn_mode))
    quant = None if args.quant == "none" else args.quant
    bs, seq = args.batch, args.seq_len
    total_steps = args.steps + args.warmup + 4
    import numpy as np

    # per-request prompt lengths: uniform, or drawn like bench_serving.py:1013 draws them (seeded; every rank draws the same)
    lens = ([seq] * bs if args.seq_dist == "uniform" else
            [int(x) for x in np.random.RandomState(5).randint(args.seq_min, seq + 1, bs)])
    P = args.shared_prefix            # tokens of the common prefix (0: none)
    kv_tokens = P + sum(n - P for n in lens)   # distinct cached tokens after the prefill
    runner = SyntheticModelRunner(cfg, quant, max_running_requests=bs, context_len=max(lens) + total_steps + 8,
                                  max_total_tokens=kv_tokens + (2 * seq if P else 0) + bs * total_steps + 64, device=dev, seed=0,
                                  dtype=torch.float16 if args.dtype == "f16" else torch.bfloat16,
                                  kv_cache_dtype=torch.float8_e4m3fn if args.kv_cache_dtype == "fp8_e4m3" else None,
                                  max_kv_splits=args.max_kv_splits, kv_split_rule=args.kv_split_rule,
                                  kv_sched_rounds_pct=args.kv_sched_rounds_pct)
    if args.no_pad_qkv:
        runner.model.pad_qkv_rows = False
    if args.no_fused_decode:
        runner.model.fused_decode = False
    if args.no_graph_metadata:
        runner.graph_metadata = False
    kv_es = 1 if args.kv_cache_dtype == "fp8_e4m3" else 2
    tp = comm.get_tensor_model_parallel_world_size()

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    phase("setup")
    # ---- prefill: fills the KV pool (random token ids as bench_one_batch.py:215) and measures prefill TFLOP/s ----
    # (ids below 10 000 as bench_one_batch.py:215 draws them, and below the vocabulary: `--model tiny` has 2 048 rows -- round 4
    # found its lookups reading 8 MB past the embedding table, garbage that ended in a GPU fault one step later)
    rs = np.random.RandomState(0)
    vmax = min(10000, cfg.vocab_size)
    if P:
        common = rs.randint(0, vmax, P)
        prompts = [torch.from_numpy(np.concatenate([common, rs.randint(0, vmax, n - P)])).to(dev) for n in lens]
    else:
        prompts = [torch.from_numpy(rs.randint(0, vmax, n)).to(dev) for n in lens]

    def prefill_all():
        """The whole batch's prompts -> KV pool; returns (states, last-token logits) in request order.  With a shared prefix: the
        RadixAttention hit path -- request 0 in full, its slots inserted into the radix tree, every other request matched against the
        tree and extended over the cached prefix (scheduler flow of schedule_batch.py / radix_cache.py: match_prefix -> extend)."""
        states, logits = [], []
        if not P:
            for c0 in range(0, bs, args.prefill_chunk):
                lg, st = runner.extend(prompts[c0:c0 + args.prefill_chunk])
                states.append(st)
                logits.append(lg)
            return states, logits
        from ltp_sglang_amd.srt.mem_cache.radix_cache import RadixCache

        cache = RadixCache(runner.req_to_token_pool, runner.token_to_kv_pool_allocator, page_size=1)
        lg, st = runner.extend([prompts[0]])
        cache.insert(prompts[0].tolist(), runner.req_to_token_pool.req_to_token[st.req_pool_indices[0], : lens[0]].to(torch.int64))
        states.append(st)
        logits.append(lg)
        for c0 in range(1, bs, args.prefill_chunk):
            chunk = prompts[c0:c0 + args.prefill_chunk]
            hits = [cache.match_prefix(p_.tolist()).device_indices for p_ in chunk]
            if not all(int(h.numel()) == P for h in hits):
                raise SystemExit(f"[bench] rank {rank}: the radix cache matched {[int(h.numel()) for h in hits][:4]}... tokens, expected the {P}-token prefix")
            lg, st = runner.extend([p_[P:] for p_ in chunk], prefix_indices=[h.to(dev) for h in hits])
            states.append(st)
            logits.append(lg)
        return states, logits

    # untimed warm-up prefill, undone: the SAME calls at full length (bench_one_batch.py warms up with the batch it then measures) --
    # it loads every prefill kernel's code object, sets the LDS attributes, and lets the caching allocator obtain the multi-GB
    # activation buffers once.  (Round 4: with a 512-token warm-up the timed prefill made those first hipMallocs itself; one run of
    # the round read 0.677 s where the runs around it read 0.46 s.)
    prefill_all()
    torch.cuda.synchronize()
    runner.clear()
    # the allocator hands out consecutive slots; a random permutation of the free list makes the gather non-contiguous
    alloc = runner.token_to_kv_pool_allocator
    gperm = torch.Generator(device=dev).manual_seed(1)
    alloc.free_pages = alloc.free_pages[torch.randperm(alloc.free_pages.numel(), generator=gperm, device=dev)]
    barrier()
    phase("prefill_warmup")
    t0 = time.perf_counter()
    states, last_logits = prefill_all()
    barrier()
    prefill_s = time.perf_counter() - t0
    phase("prefill")
    from types import SimpleNamespace

    state = SimpleNamespace(req_pool_indices=torch.cat([s.req_pool_indices for s in states]),
                            seq_lens=torch.cat([s.seq_lens for s in states]),
                            seq_lens_cpu=sum([s.seq_lens_cpu for s in states], []))
    L, hid, inter, V = cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    hq, hkv, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    # Bytes and FLOPs of what the timed processes execute: per-rank shard figures (heads / columns / rows / vocab divided by tp, kv
    # heads replicated when there are fewer than tp: llama.py:118-133) times the number of ranks that really ran -- `world` real
    # ranks, or ONE under --emulate-tp (round 2 divided the unsharded figures by a shard's time there: a "fraction" of 1.96).
    hq_s, hkv_s = hq // tp, max(1, hkv // tp)
    ranks_run = 1 if args.emulate_tp > 1 else world
    lin_params = ranks_run * L * ((hq_s * d + 2 * hkv_s * d) * hid + hq_s * d * hid + 3 * (inter // tp) * hid)
    # what the timed prefill executes: the linears over the NEW tokens of every call, causal attention of the new tokens over
    # (cached prefix + themselves), the lm_head on each request's last token
    new_tok = [lens[0]] + [n - P for n in lens[1:]] if P else list(lens)
    pre_tok = [0] + [P] * (bs - 1) if P else [0] * bs
    attn_flops = sum(4.0 * hq_s * d * (nw * pr + nw * nw / 2.0) for nw, pr in zip(new_tok, pre_tok))
    prefill_flops = 2.0 * lin_params * sum(new_tok) + ranks_run * (L * attn_flops + 2.0 * (V // tp) * hid * bs)
    first_logits = torch.cat(last_logits)
    if not bool(torch.isfinite(first_logits.float()).all()):   # outside the timed regions; a throughput over NaNs is not a measurement
        raise SystemExit(f"[bench] rank {rank}: the prefill produced non-finite logits")
    next_ids = K_argmax(first_logits)

    # ---- decode: W warm-up + K timed steps ----
    use_graph = not args.no_graph
    if use_graph and world > 1 and args.dist_backend == "gloo" and comm._CUSTOM_AR is None:
        # gloo rehearsal without the P2P communicator: the collectives are host copies, which no graph can hold -- eager by design
        use_graph = False
        if rank == 0:
            print("[bench] gloo rehearsal with host-staged collectives: eager steps (not the metric)", file=sys.stderr)
    graph_wanted = use_graph
    if use_graph:
        why = ""
        try:
            runner.capture_decode_graph(bs, shared_prefix_len=P)
        except Exception as e:
            why = f"{type(e).__name__}: {e}"
            use_graph = False
        if world > 1:
            # all ranks replay the graph or none does (a rank that alone fell back to eager launches would still match its peers'
            # collectives one for one, but the job's time would be that rank's): agree over the process group
            import torch.distributed as dist

            flag = torch.tensor([1 if use_graph else 0], device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_graph = bool(int(flag.item()))
        if not use_graph and rank == 0:
            # stated, not silent: config.hip_graph in the JSON line is false and the step is launch-bound (gloo rehearsal: host-staged
            # collectives cannot be captured, eager by design)
            print(f"[bench] HIP-graph capture of the decode step not used ({why or 'a peer rank could not capture'}); "
                  + ("timing eager steps" if args.allow_eager else "exit 3 (--allow-eager times launch-bound eager steps instead)"),
                  file=sys.stderr)
        if graph_wanted and not use_graph and not args.allow_eager:
            # a launch-bound eager number must never be recorded as the metric by accident: every rank leaves (they all agree)
            raise SystemExit(3)
    phase("capture")
    base_fn = runner.decode_graph if use_graph else runner.decode
    step_fn = (lambda st_, ids_: base_fn(st_, ids_, shared_prefix_len=P)) if P else base_fn
    for _ in range(args.warmup):
        next_ids = K_argmax(step_fn(state, next_ids))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next_ids = K_argmax(step_fn(state, next_ids))
    barrier()
    elapsed = time.perf_counter() - t0
    if not (int(next_ids.min()) >= 0 and int(next_ids.max()) < cfg.vocab_size):
        raise SystemExit(f"[bench] rank {rank}: the last decode step sampled token ids outside the vocabulary (non-finite logits?)")
    if world > 1 and comm._CUSTOM_AR is not None:
        comm._CUSTOM_AR.check_error()   # a peer that missed a spin bound leaves sums unreduced: fail instead of reporting a time
    try:
        runner.check_errors()   # persistent MLP launch (SGL_MI355_MLP_BLOCK=1): a hand-off that timed out voids the steps
    except RuntimeError as e:
        raise SystemExit(f"[bench] rank {rank}: {e}")
    phase("decode")
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant kernel: decode attention stage 1, HIP-event timed on the launch stream over all layers' pools ----
    from ltp_sglang_amd import sgl_kernel as K

    cur_seq = state.seq_lens_cpu[0]
    md = runner.attn_backend.forward_metadata
    hq_r, hkv_r = hq // tp, max(1, hkv // tp)
    qd = torch.randn(bs, hq_r, d, device=dev).to(runner.dtype)
    pool = runner.token_to_kv_pool
    evs = []
    be = runner.attn_backend
    fp8_lin = quant in ("w8a8_fp8", "fp8")
    in_launch_merge = d in (64, 128) and hq_r * d <= 16384 and (
        (fp8_lin and runner.model.fused_decode and runner.model.fused_attn_merge) or (not fp8_lin and be.merge_in_launch))
    kscale, vscale = runner.model.layers[0].self_attn.kv_scales()
    kscale, vscale = (float(kscale) if kscale is not None else 1.0), (float(vscale) if vscale is not None else 1.0)
    for rep in range(3):
        for l in range(L):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if md.cascade_prefix_indices is not None:
                # the launch pair the captured step runs over a shared prefix: prefix once for all requests + private suffixes, merged
                K.decode_attention_cascade(qd, pool.get_key_buffer(l), pool.get_value_buffer(l), md.cascade_prefix_indices,
                                           md.cascade_prefix_splits, md.kv_indptr, md.kv_indices, md.attn_logits, md.attn_lse,
                                           md.num_kv_splits, be.max_kv_splits, d ** -0.5, be._merge_counter_buf(qd), 0.0, kscale, vscale,
                                           want_o=not fp8_lin, want_quant=fp8_lin)
            elif in_launch_merge:
                # the launch the captured step runs: stage 1 + the stage-2 merge + per-token fp8 quant of the merged rows by each
                # request's last workgroup (r3's probe timed stage 1 alone, a cheaper launch than the step's)
                K.decode_attention_merge_quant(qd, pool.get_key_buffer(l), pool.get_value_buffer(l), md.kv_indptr, md.kv_indices,
                                               md.attn_logits, md.attn_lse, md.num_kv_splits, be.max_kv_splits, d ** -0.5,
                                               be._merge_counter_buf(qd), 0.0, kscale, vscale, want_o=not fp8_lin, want_quant=fp8_lin,
                                               sched=md.sched)
            else:
                # o = None: stage 1 only (the split partials)
                K.decode_attention_fwd(qd, pool.get_key_buffer(l), pool.get_value_buffer(l), None, md.kv_indptr, md.kv_indices,
                                       md.attn_logits, md.attn_lse, md.num_kv_splits, be.max_kv_splits, d ** -0.5)
            e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    attn_ms = sorted(a.elapsed_time(b) for a, b in evs[L:])  # first sweep = warm-up
    attn_ms = sum(attn_ms) / len(attn_ms)
    # algorithmic bytes: K and V rows of every DISTINCT cached token (a shared prefix is counted once: what a perfect kernel reads)
    kv_rows = float(sum(state.seq_lens_cpu)) - (bs - 1) * P
    kv_bytes = kv_rows * hkv_r * d * 2 * kv_es
    achieved = kv_bytes / (attn_ms * 1e-3) / 1e9

    phase("roofline_probe")
    if rank != 0:
        return
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so the per-launch figure
    # comes from the newest committed summary of a `rocprofv3 --pmc` pass over this same command (tools/profile_round.sh,
    # tools/pmc_traffic.py), together with the git SHA that profile was taken at
    traffic, traffic_src, traffic_sha = None, None, None
    if (args.model, args.quant, bs, seq, world, args.kv_cache_dtype, args.emulate_tp) == ("llama3-8b", "w8a8_fp8", 32, 2048, 1, "auto", 0):
        pmc_rel, pmc = _traffic_profile()
        k1 = (pmc or {}).get("kernels", {}).get("decode_attn_stage1", {})
        if "fetch_bytes_per_launch" in k1:
            traffic = k1["fetch_bytes_per_launch"] + k1.get("write_bytes_per_launch", 0.0)
            traffic_src = f"{pmc_rel} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; FETCH_SIZE x2 on gfx950)"
            traffic_sha = pmc.get("git_sha")
    tok_s = bs * args.steps / elapsed
    weights_bytes = lin_params * (1 if quant in ("w8a8_fp8", "fp8") else (0.5 if quant == "awq" else 2)) + ranks_run * (V // tp) * hid * 2
    step_bytes = weights_bytes + ranks_run * (kv_tokens + bs * (args.warmup + args.steps / 2)) * L * 2 * hkv_s * d * kv_es
    out = {
        "metric": ("decode tokens/sec (whole job) + prefill TFLOPS, Llama-3-8B fp8 batch=32 seq=2048"
                   if (args.model, args.quant, bs, seq, args.kv_cache_dtype, args.seq_dist, P) == ("llama3-8b", "w8a8_fp8", 32, 2048, "auto", "uniform", 0) else
                   f"decode tokens/sec (whole job) + prefill TFLOPS, {args.model} {args.quant} batch={bs} seq={seq} kv={args.kv_cache_dtype}"
                   + (f" ragged U({args.seq_min},{seq})" if args.seq_dist == "ragged" else "") + (f" shared prefix {P}" if P else ""))
                  + ("" if use_graph else " [EAGER launches, no HIP graph: launch-bound, not the metric]"),
        "value": tok_s, "unit": "tokens/s", "n_gpus": args.gpus, "world_size_observed": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "fp8_e4m3 weights+activations (f32 accumulate), bf16 KV/attention" if quant in ("w8a8_fp8", "fp8") else str(quant or "bf16"),
        "data": "synthetic",
        "config": {"workload": f"{args.model} {args.quant} decode, batch {bs} x context "
                               + (f"U({args.seq_min}, {seq}) (seeded; mean {sum(lens) / bs:.0f})" if args.seq_dist == "ragged" else f"{seq}")
                               + (f" = {P} shared (one radix node, cascade attention) + {seq - P} private" if P else "")
                               + f" (+{args.warmup}+{args.steps} steps), KV pool filled by a real prefill of the same prompts"
                               + (" through the radix hit path" if P else ""),
                   "baseline_config": args.config or None, "global_batch": bs, "seq_len": seq, "seq_dist": args.seq_dist,
                   "seq_lens": {"min": min(lens), "mean": sum(lens) / bs, "max": max(lens)}, "shared_prefix": P,
                   "parallelism": f"tp{tp}" + (" (EMULATED: rank 0's shard in one process, collectives = same-size device copies)" if args.emulate_tp > 1 else ""),
                   "all_reduce": ar_kind, "hip_graph": bool(use_graph), "layers": L, "kv_cache_dtype": args.kv_cache_dtype,
                   "act_dtype": args.dtype},
        "prefill": {"tflops": prefill_flops / prefill_s / 1e12, "seconds": prefill_s, "tokens": sum(new_tok), "prompt_tokens": sum(lens),
                    "tokens_per_s": sum(new_tok) / prefill_s, "prompt_tokens_per_s": sum(lens) / prefill_s, "flops": prefill_flops,
                    "requests_per_extend_call": min(bs, args.prefill_chunk)},
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "hbm_peak_GBps": 8000.0,
                          "ranks_counted": ranks_run,
                          "frac_of_hbm_roofline": step_bytes / (elapsed / args.steps) / (8e12 * ranks_run)},
        "roofline": {"kernel": "decode_attn_stage1" + (" (+ in-launch stage-2 merge + fp8 quant, the launch the step runs)" if in_launch_merge else " (stage 1 only)"),
                     "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                     "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                     "traffic_source": traffic_src, "traffic_profile_git_sha": traffic_sha, "launch_us": attn_ms * 1e3, "algorithmic_bytes_per_launch": kv_bytes},
    }
    # what the split rule made of the batch (SURVEY 7 hard-part (d): split-KV balance under raggedness)
    sp = md.num_kv_splits[:bs].cpu().tolist()
    out["kv_splits"] = {"rule": args.kv_split_rule, "max_kv_splits": args.max_kv_splits,
                        "histogram": {str(v): sp.count(v) for v in sorted(set(sp))}, "workgroup_units": int(sum(sp)) * hkv_r}
    if md.sched is not None:   # sorted unit list: {T, units, total tokens, capacity}
        hdr = md.sched[:4].cpu().tolist()
        out["kv_splits"].update({"split_tokens": hdr[0], "list_units": hdr[1], "list_capacity": hdr[3], "rounds_pct": args.kv_sched_rounds_pct})
    if not args.no_cpu_baseline and args.model != "tiny" and world == 1 and args.emulate_tp <= 1 and args.config in (0,):  # rank 0 at N = 1 only; the headline run
        out["cpu_baseline"] = cpu_baseline_sample(cfg, bs, seq, L)
    phase("cpu_baseline")
    out["phases_s"] = phases
    print(json.dumps(out))


def K_argmax(logits):
    from ltp_sglang_amd import sgl_kernel as K

    return K.argmax(logits)


if __name__ == "__main__":
    main()
