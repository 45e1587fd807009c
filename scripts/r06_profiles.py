"""Turn the outputs of scripts/r06_record.sh (gpurun_out/r6rec) into the tracked profile notes of round 6.
usage (repo root): python scripts/r06_profiles.py [gpurun_out/r6rec]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6rec")
P = os.path.join(ROOT, "profiles")
shutil.copy(R + "/bench_default.json", P + "/r06_bench_default.json")
shutil.copy(R + "/bench_default_detail.json", P + "/r06_bench_detail.json")
shutil.copy(R + "/trace_summary.md", P + "/r06_bench_kernel_trace.md")
c = json.load(open(R + "/bench_default.json"))
NOISE = "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n"
kt = open(R + "/kernel_table.md").read().replace(NOISE, "")
e32 = "%.1f" % (1e3 * float(re.search(r"dist_rows euclidean \| 1000000 \| 32 \| ([\d.]+)", kt).group(1)))
open(P + "/r06_kernel_table.md", "w").write('''# Round 6: device-resident throughput per streaming kernel (scripts/bench_kernels.py on one MI355X, final tree)

Every timed call reads the NEXT buffer of a rotation whose sum exceeds 640 MB (2.5 x the 256 MiB Infinity Cache); HIP events on the
library's stream, 30 launches after 3 warm-ups (as `profiles/r05_kernel_table.md`).  New this round: rows of 2 (and 4) summaries are
owned by lanes (`dist_rows_narrow_kernel`, `dist_rows_mahalanobis_narrow_kernel`, `dist_multiw_narrow_kernel`,
`adaptive_narrow_kernel`: U 16-byte non-temporal loads in flight per lane, no LDS tile, contiguous output blocks through a wave
LDS stage) instead of passing 2 KiB tiles through LDS.  The 4 10^6 x 2 block, round 5 -> round 6 (of 8 TB/s): euclidean 0.55 ->
0.72, minkowski p=3 0.23 -> 0.61, mahalanobis 0.28 -> 0.69, K-weight 0.26 -> 0.62, adaptive pass 0.16 -> 0.40 (0.126 -> 0.050
ms: one read instead of three; what is left is VALU time -- three correctly rounded square roots and two Chan updates with a
division per four rows -- beside 160 MB of traffic), the two-pass welford entry point unchanged (0.33: four launches of ~12 us;
the adaptive path no longer uses it at this width).  The 1000 smallest of 10^6 distances: 56 -> 37 us (0.18 -> 0.27: the
resident selection keeps a slice's keys in LDS, `profiles/r06_selection_timeline.md`).  Boxes of the pool differ by +-3 %%: 10^6 x 32 euclidean %s us here,
41.4-43.2 over the round.

''' % e32 + kt)
fetch = float(re.search(r"per dispatch\s+([\d.]+)", open(R + "/pmc_FETCH_SIZE.txt").read()).group(1))
write = float(re.search(r"per dispatch\s+([\d.]+)", open(R + "/pmc_WRITE_SIZE.txt").read()).group(1))
traffic = int(round(fetch * 1024 * 2 + write * 1024))
json.dump({"n": 1000000, "m": 32, "kernel": "dist_rows_dma_kernel<0,false,32,64,2>", "fetch_size_kib": fetch,
           "write_size_kib": write, "fetch_correction": 2.0, "traffic_bytes_per_launch": traffic,
           "source": "profiles/r06_distance_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, round 6)"},
          open(P + "/distance_pmc.json", "w"))
tr = open(R + "/trace_summary.md").read()
m = re.search(r"dist_rows_dma_kernel<0, false, 32, 64, 2>[^|]*\| (\d+) \| ([\d.]+) \| ([\d.]+)", tr)
calls, tot, avg = m.groups()
kms = c["roofline"]["kernel_ms"]
open(P + "/r06_distance_pmc.md", "w").write('''# rocprofv3 PMC passes for the distance kernel, round 6 (separate runs, one counter each)

Command (each): `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bolfi --no-cfg4 --e2e off`
(`scripts/r06_record.sh`).  Kernel: `elfihip::dist_rows_dma_kernel<0, false, 32, 64, 2>` (LDS-DMA row stream, the sampler's
selection fused in through `RejectFilter`; unchanged since round 5) on 10^6 x 32 f64 rows, 28 dispatches per pass.

| counter | per-dispatch value | bytes |
|---|---|---|
| FETCH_SIZE | %.1f KiB | %.2f MB raw; **x2 = %.1f MB** (gfx950 reports half the bytes of a 16 B/lane coalesced stream: MI355X_MICROARCH.md section HBM) |
| WRITE_SIZE | %.1f KiB | %.2f MB (8.00 MB of distances + the candidate list) |

HBM traffic per launch = %.1f MB against 264.0 MB algorithmic ((8 m + 8) bytes x 10^6 rows): one read, no write
amplification (%.3f x).  Kernel duration: %s us average over %s calls in the kernel trace of the same command
(`profiles/r06_bench_kernel_trace.md`), %.2f us from HIP events inside bench.py (`profiles/r06_bench_default.json`,
roofline.kernel_ms) -> 264.0 MB / %.2f us = %.2f TB/s = **%.3f of 8 TB/s** (%.2f of the guide's 6.29 TB/s copy ceiling).
`profiles/distance_pmc.json` (read by bench.py as `roofline.traffic`, `traffic_source: "static"`) is this record.
''' % (fetch, fetch * 1024 / 1e6, fetch * 1024 * 2 / 1e6, write, write * 1024 / 1e6, traffic / 1e6, traffic / 264e6, avg, calls,
       kms * 1e3, kms * 1e3, 264.0 / kms / 1e3, c["roofline"]["frac"], 264.0 / kms / 1e3 / 6.29))
txt = open(R + "/gp_pmc.txt").read()
blocks = re.split(r"^== ", txt, flags=re.M)[1:]
want = {"gram_kernel", "potf2_tiles_kernel<1024>", "trsm16_kernel", "lookahead_tile_kernel<1>", "step_kernel",
        "alpha_logdet_kernel", "kstar_kernel", "tri_apply_kernel<0, true>", "tri_apply_kernel<1, true>",
        "tri_apply_kernel<3, true>", "finish_kernel", "kinv_grad_kernel", "dense_tri_kernel<0, 32>", "dense_tri_kernel<1, 32>",
        "dense_tri_kernel<0, 16>", "dense_tri_kernel<1, 16>", "mirror_kernel"}
rows = []
for b in blocks:
    head = b.split(" :")[0]
    for km in re.finditer(r"^(\S.*?)  \((\d+) dispatches\)\n((?:    .*\n)+)", b, flags=re.M):
        name, nd, body = km.group(1), int(km.group(2)), km.group(3)
        if name not in want:
            continue
        v = {mm.group(1): float(mm.group(2)) for mm in re.finditer(r"(\w+)\s+\d+\s+per dispatch\s+([\d.]+)", body)}
        if "GRBM_GUI_ACTIVE" not in v:
            continue
        act, mf = v["GRBM_GUI_ACTIVE"] / 8, v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
        rows.append((head, name, nd, mf, act, mf / act if act else 0.0, v["SQ_BUSY_CYCLES"] / 8 / act if act else 0.0))


def util(cmd, kernel):
    for r in rows:
        if r[0] == cmd and r[1] == kernel:
            return r[5]
    return float("nan")


out = '''# rocprofv3 PMC passes for the GP kernels, round 6 (MFMA utilisation; the last such pass was round 2's)

Command (each): `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python <script>`
(`scripts/r06_record.sh`, one counter group per run, kernel trace only).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs
/ (GRBM_GUI_ACTIVE / 8 XCDs), as in `profiles/r02_gp_pmc.md`; "SQ busy" = SQ_BUSY_CYCLES / 8 / (GRBM_GUI_ACTIVE / 8) (the counter
sums the shader engines' busy cycles of an XCD: values above 1 mean several engines were busy).

| command | kernel | dispatches | MFMA busy / SIMD (cycles) | active cycles / XCD | MFMA utilisation | SQ busy |
|---|---|---|---|---|---|---|
'''
for r in rows:
    out += "| `%s` | `%s` | %d | %.0f | %.0f | %.2f | %.2f |\n" % r
out += '''
Reading:
* `step_kernel` (diagonal block k+1 on workgroup 0 beside the levelled trailing update on 255 workgroups), n = 4096: the matrix
  pipes are busy **%.2f** of the launch (round 2: 0.41 -- the step has not changed since round 3), %.2f at n = 2048, where the launch
  is as long as its diagonal block (27-31 us of dependent 16-column panels on ONE workgroup).  Together with the two chain launches
  between steps (6.5 + 5.3 us + three boundaries, matrix pipes idle) that is the rebuild's 0.34-0.36 of the FP64-matrix peak on
  executed flops -- the floor stated in DESIGN.md section 7.2.
* `tri_apply_kernel<0/1/3>` (the lock-step's streaming products, 16 query columns): %.2f-%.2f (%.2f for the K^-1 product) -- on this
  part the f64 matrix rate equals the f64 vector rate, so 16 columns per 8 bytes of the factor keep the pipes busy an eighth of
  the time the bytes take; their roofline is bytes (profiles/r06_lockstep_timeline.md).
* `dense_tri_kernel<*, 32>` (configs[4]: 256 starts, n = 8192): **%.2f-%.2f** of the matrix pipes while they run; the acquisition's
  0.31 of peak as a whole is these rounds + the straggler tail of streaming rounds (scripts/cfg5_trace.py: 24 rounds at 256 points,
  0.75-0.85 ms each, then ~45 rounds with fewer than 60 live starts at 0.13-0.28 ms) + 7 ms of host state machines.
* `kinv_grad_kernel` (K^-1 = L^-T L^-1 with the gradient contractions fused, one MAP-search gradient): %.2f at n = 4096, %.2f at 8192.
''' % (util("scripts/fit_once.py 4096 10 2 5", "step_kernel"), util("scripts/fit_once.py 2048 10 2 5", "step_kernel"),
       util("scripts/lcb_loop.py 4096 10", "tri_apply_kernel<1, true>"), util("scripts/lcb_loop.py 4096 10", "tri_apply_kernel<0, true>"),
       util("scripts/lcb_loop.py 4096 10", "tri_apply_kernel<3, true>"),
       util("scripts/cfg5_trace.py", "dense_tri_kernel<1, 32>"), util("scripts/cfg5_trace.py", "dense_tri_kernel<0, 32>"),
       util("scripts/lcb_loop.py 4096 10", "kinv_grad_kernel"), util("scripts/cfg5_trace.py", "kinv_grad_kernel"))
open(P + "/r06_gp_pmc.md", "w").write(out)
print("profiles written; headline", c["value"], "frac", c["roofline"]["frac"], "bolfi", c["roofline"]["bolfi_iters_per_s"])
