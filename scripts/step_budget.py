# coding: utf-8
"""Lower-bound budget of the training step as THIS decomposition launches it.

For every kernel class of the step (BASELINE configs[1]: B=64 x (64+64) tokens, Transformer-base, V=32000):

    floor = max(FLOPs / 2.5 PF, algorithmic HBM bytes / 6.3 TB/s, L2->LDS staged bytes / 12 TB/s) + 1.5 us fixed

per launch (the formula of VERDICT r02 item 1a; 6.3 TB/s = measured copy rate, 12 TB/s = the chip-wide L2 -> LDS
rate the GEMM K loops sustain), summed over the launches of a step.  Round 4: the fixed cost is
the MEASURED dependent kernel boundary of /opt/skills/guides/MI355X_MICROARCH.md -- 1.1-1.9 us, 1.5 us here -- and no
longer the 5 us rounds 2-3 used, which had priced ring fill, cold first loads and drain of every short kernel as if
they were a law of the chip.  They are a property of the decomposition (222 launches that each restart their
pipelines); the table shows them as `excess` = measured - floor per class, the prize for fewer, longer launches.
--fixed-us 5 reproduces the old table.  The GEMM tiles are the ones the library picks (zk_gemm_plan: a host function, no GPU needed); the staged
bytes of a tile grid are tiles x (BM + BN) x K x 2.  `measured` columns come from a rocprof summary
(profiles/*_rocprof_kernel_stats_*.txt) when one is given.

    python scripts/step_budget.py [profiles/r02_rocprof_kernel_stats_final.txt] [--sentences 64]
"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from zero_amd import hip  # noqa: E402

PF, HBM, L2LDS, FIXED = 2.5e15, 6.3e12, 12e12, 1.5e-6


def plan(M, N, K, out_f32=0, plain=0):
    code = hip.lib().raw("zk_gemm_plan")(M, N, K, out_f32, plain)
    return ((code >> 8) & 255) * 8, ((code >> 16) & 255) * 8, max(1, (code >> 24) & 15)


class Budget(object):
    def __init__(self):
        self.rows = collections.OrderedDict()

    def add(self, cls, n, flops=0.0, hbm=0.0, staged=0.0):
        """n launches, each with these per-launch figures."""
        t = max(flops / PF, hbm / HBM, staged / L2LDS) + FIXED
        r = self.rows.setdefault(cls, {"n": 0, "flops": 0.0, "hbm": 0.0, "staged": 0.0, "floor": 0.0, "fixed": 0.0})
        r["n"] += n
        r["flops"] += n * flops
        r["hbm"] += n * hbm
        r["staged"] += n * staged
        r["floor"] += n * t
        r["fixed"] += n * FIXED

    def gemm(self, cls, n, M, N, K, out_bytes=2, extra_in=0.0, tile=None, splits=None, extra_flops=0.0):
        bm, bn, s = plan(M, N, K, 1 if out_bytes == 4 else 0)
        if tile is not None:
            bm, bn = tile
        if splits is not None:
            s = splits
        tiles = -(-M // bm) * -(-N // bn)
        staged = tiles * (bm + bn) * K * 2.0
        hbm = (M * K + K * N) * 2.0 + M * N * out_bytes + extra_in
        self.add(cls, n, 2.0 * M * N * K + extra_flops, hbm, staged)


def build(B, L=64, H=512, F=2048, V=32000, NE=6, ND=6, nh=8, sync_ln=True):
    """sync_ln (round 4, the default launch structure): the residual + LayerNorm forward runs in the epilogue of the
    sub-layer output products (zk_gemm_add_ln: + the residual read and the saved sum written) and its backward in the
    dgrad launch that completes its input gradient (zk_gemm_ln_bwd: + the saved sum read and dy written; the dgrad result
    itself is not written) -- 58 of the 60 LayerNorm launches are gone, the two at the top of the stacks stay."""
    T = B * L
    b = Budget()
    act = T * H * 2.0                     # one bf16 activation
    # (zk_attn_out_ln: the attention forward of a sub-layer runs in its output-projection launch: q, k, v in, att out)
    af, ab_ = (4.0 * B * L * L * H, 4 * act) if sync_ln else (0.0, 0.0)
    fw = 2 * act if sync_ln else 0.0      # forward tail: residual in, saved sum out
    bw = 2 * act if sync_ln else 0.0      # backward: saved sum in, dy out (ds takes the place of dx)
    t64 = (64, 64) if sync_ln else None
    # ---- forward + dgrad chain of 4096-row GEMMs (func.py:14-65 and its mirrors)
    c = "small GEMM chain (linear fwd + dgrad, %d-row)" % T
    for _ in range(NE):
        b.gemm(c, 1, T, 3 * H, H)                        # qkv
        b.gemm(c, 1, T, H, H, extra_in=fw + ab_, tile=t64, extra_flops=af)     # o_map (+ attention forward + residual + LayerNorm)
        b.gemm(c, 1, T, F, H)                            # ffn enlarge
        b.gemm(c, 1, T, H, F, extra_in=fw, tile=t64)     # ffn output (+ residual + LayerNorm)
        b.gemm(c, 1, T, H, 3 * H, extra_in=act + bw, tile=t64)     # d qkv -> dx (+ residual) (+ LayerNorm backward below)
        # (d o_map: inside the attention backward launch since round 3, see c5)
        b.gemm(c, 1, T, F, H, extra_in=T * F * 2.0)      # d ffn output (ReLU mask)
        b.gemm(c, 1, T, H, F, extra_in=act + bw, tile=t64)         # d ffn enlarge (+ residual) (+ LayerNorm backward below)
    for _ in range(ND):
        for (n_, k_, ln) in ((3 * H, H, 0), (H, H, 2), (H, H, 0), (H, H, 2), (F, H, 0), (H, F, 1)):      # self qkv, o, cross q, o, ffn
            b.gemm(c, 1, T, n_, k_, extra_in=(fw if ln else 0.0) + (ab_ if ln == 2 else 0.0), tile=t64 if ln else None,
                   extra_flops=af if ln == 2 else 0.0)
        for (n_, k_, ex, ln) in ((H, 3 * H, act, 1), (H, H, act, 1), (F, H, T * F * 2.0, 0), (H, F, act, 1)):     # d qkv, d q, d ffn x 2
            b.gemm(c, 1, T, n_, k_, extra_in=ex + (bw if ln else 0.0), tile=t64 if ln else None)
    # cross-attention K/V of all layers (one grouped launch), d(encoder output) (one K-segmented launch)
    c2 = "grouped K/V projections + K-segmented d(enc)"
    b.add(c2, 1, 2.0 * T * (2 * H * ND) * H, act + 2 * ND * H * H * 2 + 2 * ND * act,
          (T // 128) * (2 * ND * H // 128) * 256 * H * 2.0)
    b.add(c2, 1, 2.0 * T * H * (2 * ND * H), 2 * ND * act + 2 * ND * H * H * 2 + act,
          (T // 64) * (H // 64) * 128 * (2 * ND * H) * 2.0)
    # ---- weight gradients: ONE grouped launch of 256 x 256 tiles (fp32 out), K = T tokens
    c3 = "grouped weight gradients (one launch, 256x256 tiles, fp32 out, K = %d)" % T
    wg_enc = NE * (H * 3 * H + H * H + 2 * H * F)
    wg_dec = ND * (H * 3 * H + 5 * H * H + 2 * H * F) + V * H
    params = wg_enc + wg_dec
    # staged bytes per output element = (256 + 256) * K * 2 / (256 * 256); operands: the X [T, in] and dY [T, out] of every
    # weight, each read once (the X of the q / k / v problems of a layer are separate operands of separate problems)
    enc_io = [(H, 3 * H), (H, H), (H, F), (F, H)]
    dec_io = [(H, 3 * H), (H, H), (H, H), (H, 2 * H), (H, H), (H, F), (F, H)]
    operands = (NE * sum(i + o for i, o in enc_io) + ND * sum(i + o for i, o in dec_io) + H + V) * T * 2.0
    b.add(c3, 1, 2.0 * params * T, params * 4.0 + operands, params / (256.0 * 256) * 512 * T * 2.0)
    # ---- logits: forward GEMM (fp32 logits), dlogits x E
    c4 = "logits forward + dlogits x E"
    b.add(c4, 1, 2.0 * T * V * H, act + V * H * 2 + T * V * 4.0, (T // 256) * (V // 256) * 512 * H * 2.0)
    b.add(c4, 1, 2.0 * T * V * H, T * V * 2.0 + V * H * 2 + act, (T // 128) * (H // 256) * 384 * V * 2.0)
    # ---- attention (func.py:218-256): forward reads q, k, v, writes out; backward reads q, k, v, dO, writes dq, dk, dv
    c5 = "attention forward / backward (one (sentence, head) tile per workgroup)"
    n_att = NE + 2 * ND
    if not sync_ln:
        b.add(c5, n_att, 4.0 * B * L * L * H, 4 * act)
    # backward: + the o_map dgrad of its (sentence, head): 2 T H H FLOPs, W_o once, per workgroup 64 x H of dY and of W_o staged
    b.add(c5, n_att, 10.0 * B * L * L * H + 2.0 * T * H * H, 7 * act + H * H * 2.0, B * nh * 2 * 64 * H * 2.0)
    # ---- residual + LayerNorm
    c6 = "residual + LayerNorm forward / backward"
    n_ln = 2 * NE + 3 * ND
    if sync_ln:
        b.add(c6, 2, 0, 4 * act)          # the backward at the top of each stack (its dout comes from a 256-tile / K-segmented launch)
    else:
        b.add(c6, n_ln, 0, 4 * act)           # x, y in; out, saved sum out
        b.add(c6, n_ln, 0, 4 * act)           # dout, saved sum in; dsum (+ dy with dropout) out
    # ---- cross entropy, Adam, the rest
    b.add("cross entropy (fp32 logits in, bf16 dlogits out)", 1, 0, T * V * 6.0)
    nparam = wg_enc + wg_dec + V * H + (NE * 2 + ND * 3) * 2 * H + H
    b.add("Adam (30 B / parameter)", 1, 0, nparam * 30.0)
    b.add("embeddings, masks, loss, column / LayerNorm-parameter reductions, zero fill, norm", 18 if sync_ln else 17, 0, 12e6)
    return b, nparam


def measured(path):
    """kernel-class -> (launches/step, us/step) from a profiles/*_rocprof_kernel_stats_*.txt file."""
    out = collections.defaultdict(lambda: [0.0, 0.0])
    text = open(path).read()
    m = re.search(r"over ([0-9.]+) steps", text)
    steps = float(m.group(1)) if m else 1.0
    for line in text.splitlines():
        f = line.split(None, 6)
        if len(f) < 7 or not f[0].endswith("%") or not f[1][0].isdigit():
            continue
        name, calls, total_ms = f[6], float(f[2]), float(f[1])
        if "k_gemm_grouped256<true, false" in name or "k_gemm_grouped<128, 256" in name or "k_gemm_grouped<256" in name:
            cls = "grouped weight gradients"
        elif "k_gemm_dlds<128, 256" in name or "k_gemm_grouped256" in name or "k_splitk_reduce" in name:
            cls = "logits forward + dlogits x E"
        elif "k_gemm_grouped<128, 128" in name or "k_gemm_kseg" in name:
            cls = "grouped K/V projections + K-segmented d(enc)"
        elif "k_gemm_dlds" in name:
            cls = "small GEMM chain"
        elif "k_attn_out_ln" in name:
            cls = "small GEMM chain"
        elif "k_attn" in name:
            cls = "attention forward / backward"
        elif "k_add_ln" in name:
            cls = "residual + LayerNorm forward / backward"
        elif "k_ce_fused" in name:
            cls = "cross entropy"
        elif "k_adam" in name:
            cls = "Adam"
        elif "at::native" in name or "rocclr" in name or "k_seed_advance" in name or "k_cast" in name:
            continue                                   # warm-up / eager-pass only
        else:
            cls = "embeddings, masks, loss"
        out[cls][0] += calls / steps
        out[cls][1] += total_ms / steps * 1e3
    return out


def main():
    global FIXED
    ap = argparse.ArgumentParser()
    ap.add_argument("stats", nargs="?")
    ap.add_argument("--sentences", type=int, default=64)
    ap.add_argument("--fixed-us", type=float, default=FIXED * 1e6,
                    help="fixed cost per launch (default: the measured kernel boundary, 1.5 us; rounds 2-3 used 5)")
    ap.add_argument("--two-launch-ln", action="store_true",
                    help="the launch structure of rounds 1-3: every residual + LayerNorm forward / backward a launch of its own")
    a = ap.parse_args()
    FIXED = a.fixed_us * 1e-6
    b, nparam = build(a.sentences, sync_ln=not a.two_launch_ln)
    meas = measured(a.stats) if a.stats else {}
    print("| kernel class | launches | GFLOP | HBM MB | L2->LDS MB | fixed us | **floor us** | measured us | excess us (fill / drain / restarts) |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot = [0, 0.0, 0.0, 0.0]
    for cls, r in b.rows.items():
        m = next((v for k, v in meas.items() if cls.startswith(k)), None)
        print("| %s | %d | %.0f | %.0f | %.0f | %.0f | **%.0f** | %s | %s |" % (
            cls, r["n"], r["flops"] / 1e9, r["hbm"] / 1e6, r["staged"] / 1e6, r["fixed"] * 1e6, r["floor"] * 1e6,
            ("%.0f (%d launches)" % (m[1], round(m[0]))) if m else "-",
            ("%.0f (%.1f per launch)" % (m[1] - r["floor"] * 1e6, (m[1] - r["floor"] * 1e6) / max(round(m[0]), 1))) if m else "-"))
        tot[0] += r["n"]; tot[1] += r["flops"]; tot[2] += r["floor"]; tot[3] += m[1] if m else 0.0
    print("| **step** | %d | %.0f | | | %.0f | **%.0f** | %s | %s |" % (
        tot[0], tot[1] / 1e9, tot[0] * FIXED * 1e6, tot[2] * 1e6, ("%.0f" % tot[3]) if meas else "-",
        ("%.0f" % (tot[3] - tot[2] * 1e6)) if meas else "-"))
    print("\nparameters %.1f M; FLOPs / 2.5 PF alone = %.0f us; the 40 %% bar = %.0f us" % (
        nparam / 1e6, tot[1] / PF * 1e6, tot[1] / PF * 1e6 / 0.4))


if __name__ == "__main__":
    main()
