"""Per-rank model of BASELINE configs[3] at 8 GPUs: one 1.25 M-row bf16 shard of the 10 M-row gallery on
ONE GPU with the two-phase protocol and its RCCL collectives on a one-rank group (`bench.py --force-protocol`, synchronous and
`--pipeline`), against the whole 10 M-row gallery on the same GPU.  Predicted strong-scaling efficiency of configs[3] at 8 ranks
= step(10 M rows, 1 GPU) / (8 x step(1.25 M-row shard)); what a one-rank group cannot show is the latency of the two all-gathers
between 8 real ranks (400 KiB and 1.6 MiB per rank and batch).  Usage: python scripts/shard_model_10m.py"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--scale-10m", "off",
                        "--image-dtype", "bf16", "--steps", "40", "--warmup", "5", *extra], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-2000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


full = bench("--workload", "10m", "--steps", "10", "--warmup", "2", "--async-tail", "0")
print("10 M rows, one GPU (bf16 image, synchronous): %.3f ms per 1024-query batch, scoring launches %.3f of the step, %.3f of the MFMA peak"
      % (full["ms_per_step"], full["roofline"]["kernel_share_of_step"], full["roofline"]["frac"]))
for ranks in (8, 4, 2):
    rows = -(-10_000_000 // ranks)
    for mode, extra in (("synchronous", ()), ("pipelined collectives", ("--pipeline",))):
        d = bench("--rows", str(rows), "--force-protocol", "--option", "rescore_grid_x=%d" % max(8, 96 // ranks), *extra)
        eff = full["ms_per_step"] / (ranks * d["ms_per_step"])
        ph = d.get("protocol_phases_ms_max_over_ranks")
        print("shard of %d ranks (%d rows), %s: %.3f ms per batch -> predicted efficiency %.3f; scoring %.3f of the step%s"
              % (ranks, rows, mode, d["ms_per_step"], eff, d["roofline"]["kernel_share_of_step"],
                 ("; stages (ms): " + ", ".join("%s %.3f" % kv for kv in ph.items())) if ph else ""))

# The same with the candidates of a query spread over the shards as in a real run (a one-rank group keeps all ~358 candidate
# rows of a query local; a rank of 8 re-scores an eighth of them): G shards of the 10 M-row gallery on this one GPU, shard 0's
# phase 1 / K-th of the gathered lists / phase 2 / merge timed with events (scripts/shard_step_model.py; no collectives)
for ranks in (8, 4, 2):
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "shard_step_model.py"), str(ranks), "10000000",
                        "image_dtype=0", "rescore_grid_x=%d" % max(8, 96 // ranks)], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("rank 0 per batch")]
    if not line:
        print("shard_step_model %d failed: %s" % (ranks, r.stderr[-500:]))
        continue
    total = float(line[0].split("=")[-1].split("ms")[0])
    print("%d shards of the 10 M gallery on one GPU: %s -> predicted efficiency %.3f before collective latency"
          % (ranks, line[0], full["ms_per_step"] / (ranks * total)))
