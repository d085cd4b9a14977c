"""What is resident on the GPU, when: classify every instant of a rocprofv3 kernel trace of the overlapped bench
(tools/gpu_r4k.sh) by the kernels running at that instant -- encoder-class kernels (each fills the chip by itself), the decode
cross-attention (HBM-bound, one workgroup per CU), only small decode kernels, nothing -- and report, per lane (queue), how
long a decode step takes under that load.  Input: the *_kernel_trace.csv of rocprofv3 --kernel-trace."""
import csv
import collections
import re
import sys


def cls(name):
    n = name
    if "cross_absorbed" in n or "dec_cross_attention" in n:
        return "xattn"
    if any(k in n for k in ("gemm_tiled", "gemm_astat", "mlp_fused", "panel_gemm", "enc_attention", "layernorm_kernel", "groupnorm", "pack_audio", "gn_fold", "row_meta")):
        return "enc"
    return "small"


def main(path):
    ev = []
    rows = []
    for r in csv.DictReader(open(path)):
        name = re.sub(r"msh::\(anonymous namespace\)::", "", r["Kernel_Name"])
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows.append((s, e, name, r.get("Queue_Id", "0")))
    rows.sort()
    # the overlapped phase = the longest run of milliseconds in which kernels of >= 3 different queues start; its middle 80 %
    t0 = rows[0][0]
    qs = collections.defaultdict(set)
    for s, e, name, q in rows:
        qs[(s - t0) // 1000000].add(q)
    multi = sorted(k for k, v in qs.items() if len(v) >= 3)
    best, cur_run = (0, 0), None
    for k in multi:
        if cur_run is not None and k - cur_run[1] <= 2:
            cur_run = (cur_run[0], k)
        else:
            cur_run = (k, k)
        if cur_run[1] - cur_run[0] > best[1] - best[0]:
            best = cur_run
    w0, w1 = t0 + best[0] * 1000000, t0 + (best[1] + 1) * 1000000
    a, b = w0 + (w1 - w0) * 0.1, w1 - (w1 - w0) * 0.1
    print("overlapped window: %.1f ms long" % ((w1 - w0) / 1e6))
    for s, e, name, q in rows:
        if e < a or s > b:
            continue
        c = cls(name)
        ev.append((max(s, a), 1, c))
        ev.append((min(e, b), -1, c))
    ev.sort()
    cur = collections.Counter()
    last = a
    acc = collections.Counter()
    for t, d, c in ev:
        dt = t - last
        if dt > 0:
            if cur["enc"] > 0 and cur["xattn"] > 0:
                k = "enc + xattn"
            elif cur["enc"] > 0:
                k = "enc (+small)" if cur["small"] else "enc only"
            elif cur["xattn"] > 1:
                k = ">=2 xattn"
            elif cur["xattn"] == 1:
                k = "1 xattn (+small)" if cur["small"] else "1 xattn only"
            elif cur["small"] > 0:
                k = "small only x%d" % min(cur["small"], 4)
            else:
                k = "idle"
            acc[k] += dt
            last = t
        cur[c] += d
    tot = sum(acc.values())
    print("share of wall time by what is resident (middle 80 %% of that window, %.1f ms):" % (tot / 1e6))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-20s %5.1f %%" % (k, 100.0 * v / tot))
    # per-class kernel-time sums and the stretch of the attention kernel
    dur = collections.defaultdict(list)
    for s, e, name, q in rows:
        if s >= a and e <= b:
            dur[name.split("<")[0].split("(")[0][:48]].append(e - s)
    print("kernel durations under load (us): count, mean, p50")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print("  %-48s %6d %8.2f %8.2f   total %.1f ms" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, sum(v) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
