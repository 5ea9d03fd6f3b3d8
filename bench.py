#!/usr/bin/env python
"""Headline benchmark: train-step images/sec, VGG16-NetVLAD + soft contrastive (wms) loss,
640x480, on N MI355X (BASELINE.json metric; workload = configs[1], per-GPU batch 24).

One step = VGG16 forward (hand-written HIP convolutions for every layer, bf16 channels-last; no
library convolution is left in the step) -> NetVLAD (HIP) -> [all-gather of the embeddings when N > 1] -> wms loss
forward+backward (HIP) -> NetVLAD backward (HIP) -> VGG backward (HIP) -> [bucketed gradient
all-reduce] -> Adam update.
Inputs are synthetic and resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5      (starts its own 8 rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

    python bench.py --workload retrieval --gpus N       (configs[4]: sharded exact top-25)

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     live HIP-event timing of the dominant hand-written kernel (scl_prof_* sink in
               the C library), priced on its governing roofline.  The events bracket every
               launch and cost ~6 % of a step, so they run over a few extra steps right after
               the K timed ones (`kernel_timing` says how many; SCL_BENCH_EVENTS_IN_TIMED_REGION=1
               moves them into the timed region).  `roofline.sustained` (N=1): what THIS device
               sustains on a bare LDS-fed bf16 MFMA loop at its power cap, timed in this process
               (scl_calibrate_mfma_bf16) — `frac` is against the datasheet peak, `frac_of_sustained`
               against that
  kernels      the same for every hand-written kernel on the path
  cpu_baseline the CPU restatement of the same step timed on the host cores (N=1 only)
  roofline_netvlad_stage   the NetVLAD head as ONE stage: SURVEY §8(d)'s algorithmic bytes and
               flops of the forward / backward divided by the SUM of its kernels' durations
  retrieval    configs[4] on this GPU (N=1): 100 k references x 10 k queries x 256, exact top-25
               with the certificate, both scoring modes, queries/s and the scan kernel's roofline
  comm         (N>1) world size as the process group reports it, per-step medians of the
               embedding all-gather and of the wait behind the last gradient all-reduce, the
               step time with 0 and 8 CUs left free by the persistent grids
  switches     every SCL_* environment switch and A/B flag this run was started with
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (--dtype f32 only: the float32 mode runs library convolutions, and MIOpen's find step would
# otherwise also time its naive reference solvers — seconds per call at this size)
for _k in ('FWD', 'BWD', 'WRW'):
    os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_' + _k, '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3     # f32 vector == f32-input MFMA
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA (no sparsity)

D, K, E = 512, 64, 32768


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=24, help='images per GPU')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'], help='backbone dtype')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--miopen-find', type=int, default=1,
                    help='--dtype f32 only (the float32 mode runs library convolutions): 1 lets MIOpen '
                         'benchmark its solvers per shape during the warm-up.  The bf16 step has no '
                         'library convolution')
    ap.add_argument('--fused-relu', type=int, default=1,
                    help='1 (default): bias / ReLU / pooling in the convolution epilogues and fused HIP '
                         'passes (csrc/vgg_glue.hip); 0: the plain PyTorch composition (A/B only)')
    ap.add_argument('--cpu-images', type=int, default=2, help='images in the CPU-baseline sample')
    ap.add_argument('--graph', type=int, default=0,
                    help='1: capture one train step (forward, backward, Adam) in a HIP graph after '
                         'the warm-up and replay it for the timed steps; 0 (default): eager. '
                         'Measured on MI355X: 14.234 ms per step either way — the host runs ahead '
                         'of the device, the step is not launch-bound')
    ap.add_argument('--side-wrw', type=int, default=-1,
                    help='weight-gradient kernels on a second HIP stream next to the backward-data '
                         'kernels: 1 on, 0 off (A/B), -1 (default) decided on this device during the '
                         'warm-up, at steady state, by timing 5 steps per setting three times (nets.autotune_side_wrw; the '
                         'setting that wins differs from box to box, the gradients do not)')
    ap.add_argument('--split-fwd', type=int, default=0,
                    help='1: the backbone forward as two half-batches pipelined on two HIP streams '
                         '(measured: forward 4.29 -> 3.97 ms, whole step within noise); 0 (default)')
    ap.add_argument('--variant', type=int, default=0,
                    help='DIAGNOSTICS: scl_debug_set_variant value for A/B runs of kernel variants '
                         'on one box (0 = production; anything else is not a benchmark result)')
    ap.add_argument('--workload', default='train', choices=['train', 'retrieval'],
                    help="train (default): the BASELINE metric's train step; retrieval: configs[4], "
                         "exact L2 top-25 of 10 k queries in 100 k references x 256, the reference "
                         "set sharded over the ranks (evaluation/top-n.py:103-106)")
    ap.add_argument('--refs', type=int, default=100000, help='retrieval: references in total')
    ap.add_argument('--queries', type=int, default=10000, help='retrieval: queries (replicated)')
    ap.add_argument('--score', default='f32', choices=['f32', 'bf16x3'],
                    help='retrieval: how candidates are nominated before the float64 re-rank')
    ap.add_argument('--n1-ref', type=float, default=0.0,
                    help="N>1: this workload's value measured at --gpus 1 (same box family); the line "
                         'then carries scaling_efficiency = value / (N * n1_ref)')
    ap.add_argument('--no-retrieval', action='store_true', help='skip the retrieval object (N=1)')
    ap.add_argument('--no-batch-sweep', action='store_true', help='skip the batch_sweep object (N=1)')
    ap.add_argument('--no-telemetry', action='store_true', help='skip the clock / power sample (N=1)')
    ap.add_argument('--force-dist', action='store_true',
                    help='--gpus 1 only: form a ONE-rank RCCL ("nccl") process group and run the '
                         'data-parallel step through it (embedding all-gather, own-rows backward, '
                         'bucketed asynchronous gradient all-reduce) — the collective path on one GPU')
    ap.add_argument('--dist-timeout', type=float, default=300.0,
                    help='seconds: rendezvous and every collective (a missing rank fails the job '
                         'after this long instead of hanging it)')
    ap.add_argument('--stub-cpu', action='store_true',
                    help='TEST ONLY: gloo on CPU with a trivial stand-in step; exercises the '
                         'launcher, the barriers and the max-over-ranks timing, measures nothing')
    return ap.parse_args()


# kernels of the NetVLAD head, by pass (names = the kernels' own names, as rocprofv3 prints them)
NETVLAD_FWD = ('vlad_split_w_kernel', 'vlad_planes_kernel', 'vlad_fwd_kernel<true>', 'vlad_fwd_kernel<false>',
               'vlad_fwd8_kernel<true>', 'vlad_fwd8_kernel<false>', 'vlad_fwd8_kernel<finish>',
               'vlad_finish_kernel', 'vlad_finish_sum_kernel', 'finish_norm_kernel', 'transpose_w_kernel',
               'rowtile16_kernel<ASSIGN>', 'aggregate_kernel<float>', 'aggregate_kernel<bf16>',
               'finish_sum_kernel')
NETVLAD_BWD = ('vlad_bwd_prologue_kernel', 'bwd_dots_kernel', 'bwd_du_kernel', 'vlad_bwd_kernel', 'vlad_bwd8_kernel',
               'vlad_dx_kernel',
               'vlad_wgrad_partial_kernel', 'vlad_wgrad_finish_kernel', 'rowtile16_kernel<DASSIGN>',
               'dx16_kernel<float>', 'dx16_kernel<bf16>', 'wgrad_finish_kernel')


def kernel_models(b, n, gb, x_bytes, slices=10):
    """Algorithmic flops / bytes per LAUNCH of each hand-written kernel.
    b = images on this GPU, n = locations per image, gb = global batch, slices = location slices
    per image of the fused NetVLAD kernels (10 at 24 x 1200 on 256 CUs)."""
    bn = b * n
    slab = b * slices * D * K * 4
    # bf16 feature maps: the float32 operands of the NetVLAD contractions are split into bf16
    # planes (two in the fused assign / aggregate kernels: 2 products per algorithmic one; two
    # per side in the grad_x kernel: 3 products) -> priced against the bf16 dense peak with the
    # flops they execute
    b2 = dict(peak_tflops=PEAK_BF16_TFLOPS, exec_mult=2.0)
    b3 = dict(peak_tflops=PEAK_BF16_TFLOPS, exec_mult=3.0)
    return {
        # fused soft-assignment + aggregation of one (image, slice): x once; a, logits, rn and
        # the slices' slabs out
        'vlad_fwd_kernel<true>': dict(flops=4.0 * bn * D * K,
                                      bytes=bn * D * 2 + bn * K * 8 + bn * 4 + slab, **b2),
        'vlad_fwd_kernel<false>': dict(flops=4.0 * bn * D * K, bytes=bn * D * 2 + slab, **b2),
        # round 4: the same two kernels with eight waves per workgroup (round 6: the logits are no
        # longer saved — a and rn only — and the backward pass reads a, not a + logits)
        'vlad_fwd8_kernel<true>': dict(flops=4.0 * bn * D * K,
                                       bytes=bn * D * 2 + bn * K * 4 + bn * 4 + slab, **b2),
        'vlad_fwd8_kernel<false>': dict(flops=4.0 * bn * D * K, bytes=bn * D * 2 + slab, **b2),
        'vlad_fwd8_kernel<stamps>': dict(flops=4.0 * bn * D * K,
                                         bytes=bn * D * 2 + bn * K * 4 + bn * 4 + slab, **b2),
        # round 6, inference at >= 144 images: one workgroup per image, the finish in its tail
        'vlad_fwd8_kernel<finish>': dict(flops=4.0 * bn * D * K + 4.0 * b * D * K,
                                         bytes=bn * D * 2 + b * D * K * 4 + 2 * D * K * 4, **b2),
        'vlad_bwd8_kernel': dict(flops=4.0 * bn * D * K,
                                 bytes=bn * D * 2 + bn * K * 8 + bn * 8 + slab + b * D * K * 4, **b2),
        # fused x.dU + softmax backward + x^T.(ds rn): x, a, logits, rn in; ds, rowdot, slabs out
        'vlad_bwd_kernel': dict(flops=4.0 * bn * D * K,
                                bytes=bn * D * 2 + bn * K * 12 + bn * 8 + slab + b * D * K * 4, **b2),
        # [a|ds].[dU|W]^T then the norm Jacobian: x, a, ds in, grad_x out, the operand images (two
        # bf16 planes of dU per image and of W; the other slices of an image find them in the L2)
        # (round 4: + the parameter-gradient sums in its tail: every dW slab once, dU once)
        'vlad_dx_kernel': dict(flops=4.0 * bn * D * K,
                               bytes=4 * bn * D + bn * K * 8 + (b + 1) * D * K * 4 + slab + b * D * K * 4,
                               **b3),
        # round 4: finish_sum + finish_norm, and bwd_dots + bwd_du, as one launch each
        'vlad_finish_kernel': dict(flops=(slices + 4.0) * b * D * K, bytes=slab + b * D * K * 8),
        'vlad_bwd_prologue_kernel': dict(flops=12.0 * b * D * K, bytes=b * D * K * 4 * 5),
        'vlad_planes_kernel': dict(flops=0.0, bytes=D * K * 12),
        'vlad_wgrad_partial_kernel': dict(flops=b * slices * D * K, bytes=slab),
        'vlad_wgrad_finish_kernel': dict(flops=4.0 * b * D * K, bytes=b * D * K * 4 + 9 * D * K * 4),
        'vlad_split_w_kernel': dict(flops=0.0, bytes=D * K * 8),
        'vlad_finish_sum_kernel': dict(flops=(slices + 2.0) * b * D * K, bytes=slab + b * D * K * 8),
        'finish_norm_kernel': dict(flops=2.0 * b * D * K, bytes=b * D * K * 4 * 2),
        'bwd_dots_kernel': dict(flops=8.0 * b * D * K, bytes=b * D * K * 4 * 2),
        'bwd_du_kernel': dict(flops=4.0 * b * D * K, bytes=b * D * K * 4 * 5),
        # float32 feature maps (and the A/B variant 8): float32-MFMA kernels
        'rowtile16_kernel<ASSIGN>': dict(flops=2.0 * bn * D * K,
                                         bytes=bn * D * x_bytes + bn * K * 8 + bn * 4),
        'rowtile16_kernel<DASSIGN>': dict(flops=2.0 * bn * D * K,
                                          bytes=bn * D * x_bytes + bn * K * 12 + b * D * K * 4),
        'aggregate_kernel<float>': dict(flops=2.0 * bn * D * K,
                                        bytes=bn * D * 4 + bn * K * 4 + b * 4 * D * K * 4),
        'aggregate_kernel<bf16>': dict(flops=2.0 * bn * D * K,
                                       bytes=bn * D * 2 + bn * K * 4 + b * 4 * D * K * 4),
        'dx16_kernel<float>': dict(flops=4.0 * bn * D * K, bytes=2 * bn * D * 4 + bn * K * 8),
        'dx16_kernel<bf16>': dict(flops=4.0 * bn * D * K, bytes=2 * bn * D * 2 + bn * K * 8),
        'finish_sum_kernel': dict(flops=6.0 * b * D * K, bytes=b * D * K * 4 * 6),
        'wgrad_finish_kernel': dict(flops=4.0 * b * D * K, bytes=b * D * K * 12),
        'transpose_w_kernel': dict(flops=0.0, bytes=D * K * 8),
        # loss: raw Gram (upper-triangular tile pairs), reads E once
        'gram_partial_kernel': dict(flops=2.0 * gb * gb * E, bytes=gb * E * 4),
        'gram16_kernel': dict(flops=2.0 * gb * gb * E, bytes=gb * E * 4),
        # round 4, B <= 32: the finish rides in the Gram kernel's last workgroup (one launch)
        'gram16_fused_kernel': dict(flops=2.0 * gb * gb * E + 48.0 * gb * gb, bytes=gb * E * 4),
        # 64 < B <= 256: the same Gram from six bf16 plane products (float32-equivalent);
        # SURVEY §8(d) prices the B = 192 Gram on the float32 MFMA peak
        'gram16x6_kernel': dict(flops=2.0 * gb * gb * E, bytes=gb * E * 4),
        # round 6: the same products, strip-scheduled (64 < B <= 208)
        'gram16x6p_kernel': dict(flops=2.0 * gb * gb * E, bytes=gb * E * 4),
        # (diagnostic build, scl_debug_set_variant(41): the whole forward as one persistent launch)
        'gram16x6_persist_kernel': dict(flops=2.0 * gb * gb * E + 48.0 * gb * gb, bytes=gb * E * 4),
        'gram16_persist_kernel': dict(flops=2.0 * gb * gb * E + 48.0 * gb * gb, bytes=gb * E * 4),
        # everything after the Gram for B <= 32, one workgroup
        'gram_final32_kernel': dict(flops=48.0 * gb * gb, bytes=gb * gb * 16),
        # slab sums (split-K artefact: priced on the Gram matrix it produces)
        'gram_reduce_kernel': dict(flops=0.0, bytes=gb * gb * 4),
        'gram_rows_kernel': dict(flops=40.0 * gb * gb, bytes=gb * gb * 16),
        'gram_rows_wave_kernel': dict(flops=40.0 * gb * gb, bytes=gb * gb * 16),
        'gram_coef_kernel': dict(flops=8.0 * gb * gb, bytes=gb * gb * 12),
        # grad_E[own rows] = M E: reads E once, writes b rows
        'gram_bwd_kernel': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
        'gram_bwd32_kernel': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
        'gram_bwd_fast_kernel<1>': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
        'gram_bwd_fast_kernel<2>': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
        'gram_bwd_fast_kernel<4>': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
        # M E on three bf16 planes each: six products executed on the bf16 matrix cores
        'gram_bwd_planes_kernel': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4,
                                       exec_mult=6.0, peak_tflops=PEAK_BF16_TFLOPS),
        'gram_bwd_rows_kernel': dict(flops=2.0 * b * gb * E, bytes=gb * E * 4 + b * E * 4),
    }


def netvlad_stage(kernels, b, n, x_bytes, steps):
    """The NetVLAD head as ONE stage per pass, the way SURVEY §8(d) prices it: algorithmic bytes
    (forward: x once + the descriptors; backward: x twice + dV) and flops (2 contractions forward,
    4 backward; on bf16 maps each runs as two bf16 plane products -> executed = 2x, against the
    bf16 dense peak) over the SUM of its kernels' durations in one step."""
    bn = b * n
    out = {}
    for name, members, nbytes, flops in (
            ('forward', NETVLAD_FWD, bn * D * x_bytes + b * D * K * 4 + 2 * D * K * 4, 4.0 * bn * D * K),
            ('backward', NETVLAD_BWD, 2 * bn * D * x_bytes + b * D * K * 4, 8.0 * bn * D * K)):
        rows = [r for r in kernels if r['kernel'] in members]
        if not rows:
            continue
        us = sum(r['us'] * r['launches'] for r in rows) / max(steps, 1)
        us_ev = sum(r.get('us_events', r['us']) * r['launches'] for r in rows) / max(steps, 1)
        t_hbm = nbytes / (PEAK_HBM_GBPS * 1e9) * 1e6
        if x_bytes == 2:
            t_mfma = 2.0 * flops / (PEAK_BF16_TFLOPS * 1e12) * 1e6
        else:
            t_mfma = flops / (PEAK_F32_TFLOPS * 1e12) * 1e6
        bound_us = max(t_hbm, t_mfma)
        out[name] = dict(kernels=[r['kernel'] for r in rows],
                         launches_per_step=round(sum(r['launches'] for r in rows) / max(steps, 1), 2),
                         us_per_step=round(us, 2), algorithmic_bytes=int(nbytes),
                         algorithmic_flops=flops, bound='hbm' if t_hbm >= t_mfma else 'mfma',
                         bound_us=round(bound_us, 2), frac=round(bound_us / us, 4) if us > 0 else None,
                         us_per_step_events=round(us_ev, 2),
                         frac_events=round(bound_us / us_ev, 4) if us_ev > 0 else None)
    out['note'] = ('durations = HIP events around every launch minus the bracket measured on the empty '
                   'kernel in this process (bracket_us in kernel_timing; *_events = the raw event figures); '
                   'rocprofv3 trace of the same command: profiles/r05')
    if x_bytes == 2:
        out['floor'] = netvlad_floor(b, n, out)
    return out


# What one CU streams from beyond its L2 with every CU streaming, measured on this chip by
# scripts/dma_line_probe.hip (profiles/r02: 12.7-13.7 B per cycle and CU from the Infinity Cache,
# 10.7 from HBM, at the 2.1 GHz the chip holds under load) — the rate the fused NetVLAD kernels'
# workgroups are bound by: bytes PER WORKGROUP, not bytes per chip, because 24 images on 256 CUs
# mean ten partial VLADs per image.
CU_STREAM_B_PER_CLK = 13.0
CU_CLOCK_GHZ = 2.1


def netvlad_floor(b, n, stage):
    """The floor of the NetVLAD head UNDER ITS DECOMPOSITION (one workgroup per CU and (image, slice
    of locations); W^T / dU^T as register fragments; one [512][64] float32 partial per workgroup):
    bytes each workgroup must move / the measured per-CU stream rate, per launch, never below the
    empty-kernel time.  SURVEY 8(d)'s HBM bound (`bound_us`) prices the bytes as if they were spread
    over the whole chip once — unreachable at 24 images, where a workgroup re-reads the 128 KB
    operand image and writes a 128 KB partial for 123 KB of x.  Reported next to the measured time:
    how much of it the decomposition explains, and how much is left to the kernels."""
    steps = -(-n // 32)
    per = -(-steps * b // 256)
    slices = -(-steps // per)
    loc = per * 32
    kb = 1024.0
    x = loc * D * 2 / kb
    frag = 2 * D * K * 2 / kb                   # two bf16 planes of a [512][64] operand: 128 KB
    slab = D * K * 4 / kb                       # a float32 partial: 128 KB
    rows = loc * K * 4 / kb                     # one saved [loc][64] float32 array
    rate = CU_STREAM_B_PER_CLK * CU_CLOCK_GHZ * 1e3          # bytes per microsecond and CU
    null_us = NULL_KERNEL_DEVICE_US or 3.4
    launches = {
        'forward': {'vlad_fwd8_kernel': x + frag + rows + slab,
                    'vlad_finish_kernel': slices * slab / 8 + 2 * slab / 8},
        'backward': {'vlad_bwd_prologue_kernel': 3 * slab / 8 + 2 * frag / 8,
                     'vlad_bwd8_kernel': x + frag + rows + rows + slab,
                     'vlad_dx_kernel': 2 * x + 2 * rows + 2 * frag + b * slices * slab / (b * slices)},
    }
    out = {'per_cu_rate_bytes_per_us': round(rate, 1), 'workgroups': b * slices, 'slices_per_image': slices,
           'how': 'sum over the launches of max(empty-kernel time, KB per workgroup / per-CU stream rate); '
                  'KB per workgroup: x slice + operand fragments + saved rows + the float32 partial'}
    for ps, ks in launches.items():
        us = sum(max(null_us, v * kb / rate) for v in ks.values())
        ent = {'kb_per_workgroup': {k: round(v, 1) for k, v in ks.items()}, 'floor_us': round(us, 2)}
        if ps in stage and stage[ps].get('us_per_step'):
            ent['floor_over_measured'] = round(us / stage[ps]['us_per_step'], 3)
        out[ps] = ent
    return out


# What the event bracket adds to a launch is measured IN THIS PROCESS (round 4 subtracted a
# constant taken from a rocprofv3 run on another box — ADVICE round 4): bracket_us() times the
# library's empty kernel (scl_null_kernel) through the bracket, and the same kernel launched 200
# times back to back between ONE event pair; the second figure per launch is the device's own
# dispatch + run time of an empty kernel, the difference is the bracket.  price() takes it off
# (`us` = corrected, `us_events` / `frac_events` = exactly as measured), never more than
# BRACKET_CAP of a duration.
NULL_KERNEL_DEVICE_US = None          # measured by bracket_us(); rocprofv3 on this chip: 3.5 us
BRACKET_US = 0.0
BRACKET_CAP = 0.3


def bracket_us(dev, n=100):
    """HIP-event bracket overhead per launch on this device, measured on the empty kernel."""
    global NULL_KERNEL_DEVICE_US
    from soft_contrastive_learning_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1 << 16, device=dev)
    st = _lib.stream_of(x)
    for _ in range(10):
        lib.scl_prof_null(st)
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=2 * n) as kt:
        for _ in range(n):
            x.add_(1.0)                       # a kernel in front, as inside a step
            lib.scl_prof_null(st)
        torch.cuda.synchronize()
    us = sorted(t * 1e3 for name, t in kt.records if name == 'scl_null_kernel')
    # the empty kernel's own time: 200 launches back to back between one event pair
    per = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.add_(1.0)
        e0.record()
        for _ in range(200):
            lib.scl_prof_null(st)
        e1.record()
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) * 1e3 / 200)
    NULL_KERNEL_DEVICE_US = sorted(per)[len(per) // 2]
    return max(0.0, us[len(us) // 2] - NULL_KERNEL_DEVICE_US) if us else 0.0


def sustained_mfma(dev, launches=8, iters=2000):
    """What this device sustains on a bare LDS-fed bf16 MFMA loop (csrc/calibrate.hip,
    scl_calibrate_mfma_bf16: one 512-thread workgroup per CU, [128 x 64] accumulators per wave, no
    global traffic, random operands), timed here — right after the steps, chip warm — with one event
    pair around `launches` back-to-back launches of ~1.4 ms each.  The datasheet figure `roofline`
    is priced against assumes 2.4 GHz; under the 1.3-1.4 kW cap the matrix cores run at 2.0-2.2 GHz,
    and boxes differ: this is the ceiling of THIS box in THIS run."""
    import ctypes
    from soft_contrastive_learning_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    ops = (torch.rand(32768, device=dev) * 2 - 1).to(torch.bfloat16)           # 64 KB
    sink = torch.zeros(cus, device=dev, dtype=torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for shape, key in ((32, 'tflops_32x32x16'), (16, 'tflops_16x16x32')):
        def go():
            rc = lib.scl_calibrate_mfma_bf16(shape, cus, iters, ctypes.c_void_p(ops.data_ptr()),
                                             ctypes.c_void_p(sink.data_ptr()), st)
            if rc != 0:
                raise RuntimeError('scl_calibrate_mfma_bf16: %d' % rc)
        for _ in range(3):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            go()
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / launches
        out[key] = round(lib.scl_calibrate_mfma_bf16_flops(cus, iters) / (ms * 1e-3) / 1e12, 1)
    out['workgroups'] = cus
    out['ms_per_launch'] = round(ms, 3)
    out['how'] = ('bare bf16 MFMA loop fed from LDS (scl_calibrate_mfma_bf16: 512 threads per CU, random '
                  'operands, no global traffic), %d launches back to back between one event pair, in this '
                  'process right after the steps' % launches)
    return out


def price(name, launches, mean_ms, model):
    events_ms = mean_ms
    mean_ms = max(mean_ms - BRACKET_US * 1e-3, (1.0 - BRACKET_CAP) * mean_ms)
    sec = mean_ms * 1e-3
    tf = model['flops'] / sec / 1e12 if sec > 0 else 0.0
    gbs = model['bytes'] / sec / 1e9 if sec > 0 else 0.0
    peak_tf = model.get('peak_tflops', PEAK_F32_TFLOPS)
    mult = model.get('exec_mult', 1.0)
    t_mfma = mult * model['flops'] / (peak_tf * 1e12)
    t_hbm = model['bytes'] / (PEAK_HBM_GBPS * 1e9)
    bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
    frac = mult * tf / peak_tf if bound == 'mfma' else gbs / PEAK_HBM_GBPS
    out = dict(kernel=name, launches=launches, us=round(mean_ms * 1e3, 2), bound=bound,
               tflops=round(tf, 2), gbps=round(gbs, 1), frac=round(frac, 4))
    if BRACKET_US:
        out['us_events'] = round(events_ms * 1e3, 2)
        out['frac_events'] = round(frac * mean_ms / events_ms, 4) if events_ms > 0 else None
    if mult != 1.0:      # bf16x3 kernels: flops executed on the bf16 matrix cores
        out.update(executed_tflops=round(mult * tf, 2), mfma_peak_tflops=peak_tf)
    return out


def cpu_baseline(args, threads):
    """The same train step restated on the CPU (oracle: torch-CPU f32 backbone + the
    autograd twin of NetVLAD and wms_loss), on a bounded sample."""
    from oracle import twin_torch as TT
    from soft_contrastive_learning_amd.model import nets
    nb = max(2, args.cpu_images)
    torch.manual_seed(0)
    model = nets.VGG16NetVLAD(compute_dtype=torch.float32)
    img = torch.randint(0, 256, (nb, args.height, args.width, 3),
                        generator=torch.Generator().manual_seed(42)).float()
    rng = np.random.default_rng(7)
    xy = rng.uniform(0, 200, size=(nb, 2))
    dist_m = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)

    def step():
        for p in model.parameters():
            p.grad = None
        fmap = model.features(img)
        emb = TT.netvlad(fmap.reshape(nb, -1, D), model.assignment_kernel.reshape(D, K),
                         model.cluster_centers.reshape(D, K), dtype=torch.float32)
        loss = TT.wms_loss(dist_m[None], emb, 0.8, 15.0, dtype=torch.float32)
        loss.backward()
        return float(loss)

    step()                                   # warm-up (allocator, oneDNN primitives)
    reps, t0 = 0, time.perf_counter()
    while True:
        step()
        reps += 1
        dt = time.perf_counter() - t0
        if reps >= 5 or dt > 30.0:             # >= 5 timed steps (30 s cap)
            break
    return dict(value=round(nb * reps / dt, 3), unit='images/sec', cores=threads, kind='port',
                sample='%d steps of %d images %dx%d fwd+bwd (torch-CPU f32 VGG16 + oracle '
                       'NetVLAD/wms autograd twin), no optimizer' % (reps, nb, args.width, args.height))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without an outer launcher: this process (which has not
    touched the GPU and never will) starts one fresh child per rank with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relays their output (rank 0 prints the JSON line), and exits
    non-zero if any rank failed."""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env))
    codes = [None] * len(procs)
    try:
        while any(c is None for c in codes):
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = pr.wait(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        pass
            if any(c not in (None, 0) for c in codes):
                break                       # one rank died: the others would hang in a collective
    finally:
        for pr in procs:                    # exactly the children started above
            if pr.poll() is None:
                pr.terminate()
        for i, pr in enumerate(procs):
            if codes[i] is None:
                try:
                    codes[i] = pr.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    codes[i] = pr.wait()
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %s\n' % bad)
        raise SystemExit(1)
    raise SystemExit(0)


def stub_main(args, world, rank):
    """--stub-cpu: the launch / barrier / timing / reporting skeleton of main() on gloo with a
    trivial step.  Exists so that the N > 1 control flow is testable without a GPU."""
    if world > 1:
        from soft_contrastive_learning_amd import parallel
        parallel.init_process_group(backend='gloo', timeout_s=args.dist_timeout)
    w = torch.ones(64, 64)

    def step():
        y = (w @ w).sum()
        if world > 1:
            dist.all_reduce(y)
        return y

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if rank == 0:
        print(json.dumps({'metric': 'STUB (no GPU work; launcher self-test)', 'value': 0.0,
                          'unit': 'none', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'none', 'data': 'stub',
                          'config': {'workload': 'stub', 'parallelism': 'dp%d' % world}}))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_kernels(threads):
    """BASELINE.md §3 / SURVEY §8d: the CPU restatement of the individual hot-path pieces on the
    host cores, each on a bounded sample — NetVLAD fused and in the reference graph's
    [N,D,K]-materialising form, the wms loss forward (float32 oracle) and forward+backward
    (autograd twin) at B = 24 and 192, and the reference's own KDTree query."""
    from oracle import losses_np as O
    from oracle import netvlad_np as NV
    from oracle import topn_np as TN
    from oracle import twin_torch as TT
    from tests import util_data as U

    def timeit(fn, budget=3.0, max_reps=20):
        fn()
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= max_reps:
                return dt / reps, reps

    out = {'cores': threads, 'kind': 'port'}
    w, c = U.vlad_params()
    x4 = U.feature_map(4, 1200, seed=5)
    t, r = timeit(lambda: NV.netvlad_fused(x4, w, c))
    out['netvlad_fwd_fused'] = dict(value=round(4 / t, 2), unit='images/sec',
                                    sample='%d x 4 images x 1200 locations, NumPy two-matmul form' % r)
    x1 = x4[:1]
    t, r = timeit(lambda: NV.netvlad_literal(x1, w, c), budget=4.0, max_reps=5)
    out['netvlad_fwd_tf_style'] = dict(value=round(1 / t, 2), unit='images/sec',
                                       sample='%d x 1 image: materialises [1200,512,64] like the '
                                              'reference graph (157 MB per image)' % r)

    def twin_step():
        xt = torch.tensor(x4[:2], requires_grad=True)
        TT.netvlad(xt, torch.tensor(w, requires_grad=True), torch.tensor(c, requires_grad=True),
                   dtype=torch.float32).sum().backward()
    t, r = timeit(twin_step)
    out['netvlad_fwd_bwd_twin'] = dict(value=round(2 / t, 2), unit='images/sec',
                                       sample='%d x 2 images, torch-CPU float32 autograd twin' % r)
    for bsz in (24, 192):
        emb = U.embeddings(bsz, E)
        dm = U.positions_distances(bsz)[None]
        t, r = timeit(lambda: O.wms_loss(dm, emb, 0.8, 15.0), budget=2.0)
        out['wms_fwd_b%d' % bsz] = dict(value=round(t * 1e6, 1), unit='us',
                                        sample='%d calls, float32 NumPy oracle' % r)

        def fb():
            e = torch.tensor(emb, requires_grad=True)
            TT.wms_loss(dm, e, 0.8, 15.0, dtype=torch.float32).backward()
        t, r = timeit(fb, budget=2.0)
        out['wms_fwd_bwd_b%d' % bsz] = dict(value=round(t * 1e6, 1), unit='us',
                                            sample='%d calls, torch-CPU float32 autograd twin' % r)
    ref, qry = U.retrieval_sets(100000, 64, 256)
    t0 = time.perf_counter()
    TN.topn_kdtree(ref, qry, 25)
    dt = time.perf_counter() - t0
    out['kdtree_100k_refs'] = dict(value=round(64 / dt, 2), unit='queries/sec',
                                   sample='KDTree(100000 x 256).query(64 queries, k=25) incl. the '
                                          'build (evaluation/top-n.py:103-106), scikit-learn')
    return out


def price_topn_scan(ms, q, r, d, score):
    """The scan kernel of csrc/topn.hip on its governing roofline: 2 Q R d algorithmic flops;
    'f32' runs them on the float32 MFMA (157.3 TF), 'bf16x3' executes THREE bf16 products per
    algorithmic one on the bf16 matrix cores (2.5 PF dense) — priced on what it executes."""
    sec = ms * 1e-3
    alg = 2.0 * q * r * d
    if score == 'bf16x3':
        ex = 3.0 * alg / sec / 1e12
        return dict(bound='mfma', tflops=round(alg / sec / 1e12, 2), executed_tflops=round(ex, 2),
                    mfma_peak_tflops=PEAK_BF16_TFLOPS, frac=round(ex / PEAK_BF16_TFLOPS, 4))
    tf = alg / sec / 1e12
    return dict(bound='mfma', tflops=round(tf, 2), mfma_peak_tflops=PEAK_F32_TFLOPS,
                frac=round(tf / PEAK_F32_TFLOPS, 4))


def retrieval_line(dev, r=100000, q=10000, d=256, n=25, iters=3):
    """BASELINE.json configs[4] on ONE GPU (evaluation/top-n.py:103-106): exact L2 top-n with the
    per-query exactness certificate, both scoring modes; whole-call queries/s (inputs resident in
    HBM, the certificate's host check included) and the scan kernel priced on its roofline."""
    from soft_contrastive_learning_amd import _lib
    from soft_contrastive_learning_amd.evaluation import retrieval
    from tests import util_data as U
    ref, qry = U.retrieval_sets(r, q, d)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    out = {'workload': 'configs[4]: %d references x %d queries x %d, exact top-%d, certified' % (r, q, d, n),
           'refs': r, 'queries': q, 'd': d, 'n': n, 'data': 'synthetic N(0,1), seeds 11 / 12'}
    for score in ('f32', 'bf16x3'):
        st = {}
        retrieval.topn_l2(rt, qt, n, score=score, stats=st)
        torch.cuda.synchronize()
        with _lib.KernelTimer(capacity=16 * iters) as kt:
            t0 = time.perf_counter()
            for _ in range(iters):
                retrieval.topn_l2(rt, qt, n, score=score, stats=st)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / iters
        summ = kt.summary()
        # round 6: the threshold scan (its 1/16 pre-pass is priced with it: 17/16 of the products)
        bfn = 1 if score == 'bf16x3' else 0
        scan_name = 'topn_scan_kernel<BF=%d,tau>' % bfn
        pre_name = 'topn_scan_kernel<BF=%d,pre>' % bfn
        if scan_name not in summ:
            scan_name, pre_name = 'topn_scan_kernel<BF=%d>' % bfn, None
        row = {'queries_per_sec': round(q / wall, 1), 'ms_per_call': round(wall * 1e3, 3),
               'uncertified_queries': st.get('uncertified'), 'calls': iters,
               'kernels_us': {k: round(ms * 1e3, 1) for k, (c, ms) in sorted(summ.items())}}
        if scan_name in summ:
            row['scan_kernel'] = dict(kernel=scan_name, us=round(summ[scan_name][1] * 1e3, 1),
                                      **price_topn_scan(summ[scan_name][1], q, r, d, score))
            if pre_name in summ:
                both = summ[scan_name][1] + summ[pre_name][1]
                row['scan_with_prepass'] = dict(us=round(both * 1e3, 1),
                                                **price_topn_scan(both, q, r * 17.0 / 16.0, d, score))
        out[score] = row
    # the in-training localisation check (train/train.py:1181-1182): the raw 32768-wide descriptors,
    # 5 nearest of a few thousand references for a few dozen queries — nomination from the inner
    # products of topn_dots_kernel (float32 matrix instructions in chunks of 256, float64 across)
    gen = torch.Generator(device=dev).manual_seed(13)
    rw, qw, dw, nw = 2000, 50, E, 5
    refw = torch.randn(rw, dw, device=dev, generator=gen)
    refw = refw / refw.norm(dim=1, keepdim=True)
    qryw = refw[torch.randperm(rw, device=dev, generator=gen)[:qw]] + 0.05 * torch.randn(
        qw, dw, device=dev, generator=gen) / dw ** 0.5
    st = {}
    retrieval.topn_l2(refw, qryw, nw, stats=st)
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=64) as kt:
        t0 = time.perf_counter()
        for _ in range(iters):
            retrieval.topn_l2(refw, qryw, nw, stats=st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
    summ = kt.summary()
    dots_ms = summ.get('topn_dots_kernel', (0, 0.0))[1]
    out['localisation_width'] = {
        'workload': 'train/train.py:1181-1182 shape: %d references x %d queries x %d (unit-norm), exact top-%d, '
                    'certified' % (rw, qw, dw, nw),
        'queries_per_sec': round(qw / wall, 1), 'ms_per_call': round(wall * 1e3, 3),
        'uncertified_queries': st.get('uncertified'),
        'dots_kernel': dict(kernel='topn_dots_kernel', us=round(max(dots_ms * 1e3 - BRACKET_US, 0.0), 1),
                            tflops=round(2.0 * rw * qw * dw / max(dots_ms * 1e-3 - BRACKET_US * 1e-6, 1e-9) / 1e12, 2),
                            peak_tflops=PEAK_F32_TFLOPS,
                            note='64 x 64 tiles x 16 feature-axis splits = 512 workgroups at this shape')}
    return out


def gpu_telemetry(step, fence, seconds=2.5):
    """Shader clock and package power UNDER THE BENCH STEP, outside the timed region: the step runs
    in a loop for `seconds` while a thread polls `rocm-smi --showclocks --showpower --json`
    (boxes of this pool differ by +-4 % in the MFMA-bound kernels — VERDICT round 4, weak item 11 —
    and a reader of BENCH_r*.json should be able to tell a slow box from a regression).  One idle
    sample first.  Never raises: a box without rocm-smi yields {'error': ...}."""
    import re
    import subprocess
    import threading

    def sample():
        try:
            r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True,
                               text=True, timeout=10)
            d = json.loads(r.stdout[r.stdout.index('{'):])
        except Exception as exc:                          # noqa: BLE001
            return {'error': '%s: %s' % (type(exc).__name__, str(exc)[:80])}
        card = d.get('card0') or next((v for v in d.values() if isinstance(v, dict)), {})
        out = {}
        for k, v in card.items():
            m = re.search(r'(\d+(?:\.\d+)?)', str(v))
            if not m:
                continue
            kl = k.lower()
            if 'sclk' in kl and 'sclk_mhz' not in out:
                out['sclk_mhz'] = float(m.group(1))
            elif 'mclk' in kl and 'mclk_mhz' not in out:
                out['mclk_mhz'] = float(m.group(1))
            elif 'power' in kl and 'power_w' not in out:
                out['power_w'] = float(m.group(1))
        return out or {'error': 'no clock / power fields in rocm-smi output', 'keys': sorted(card)[:12]}

    idle = sample()
    if 'error' in idle:
        return idle
    got, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            got.append(sample())
    th = threading.Thread(target=poll, daemon=True)
    fence()
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        step()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()                      # keep the host at most 8 steps ahead
    fence()
    stop.set()
    th.join(timeout=15)
    got = [g for g in got if 'error' not in g]

    def stat(key):
        v = sorted(g[key] for g in got if key in g)
        return {'median': v[len(v) // 2], 'min': v[0], 'max': v[-1]} if v else None
    return {'idle': idle, 'under_load': {'sclk_mhz': stat('sclk_mhz'), 'mclk_mhz': stat('mclk_mhz'),
                                         'power_w': stat('power_w'), 'samples': len(got),
                                         'steps_run': n, 'seconds': round(time.perf_counter() - t0, 2)},
            'how': 'rocm-smi polled by a thread while the bench step loops, after the timed region'}


def _switches(args):
    """Everything that can make this run differ from the default one: SCL_* environment
    switches and the A/B flags of this script."""
    sw = {k: v for k, v in sorted(os.environ.items()) if k.startswith('SCL_')}
    for name, default in (('side_wrw', -1), ('split_fwd', 0), ('variant', 0), ('graph', 0),
                          ('fused_relu', 1), ('miopen_find', 1), ('force_dist', False)):
        if getattr(args, name) != default:
            sw['--' + name.replace('_', '-')] = getattr(args, name)
    return sw


def retrieval_main(args, world, rank, dev, dp=False):
    """--workload retrieval: configs[4] with the REFERENCE SET SHARDED over the ranks (SURVEY
    §8e): every rank scans its rows for all (replicated) queries, the [Q, n] candidates are
    all-gathered and merged.  One step = one call over all queries; strong scaling (the total
    work is fixed as N grows)."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.evaluation import retrieval
    from tests import util_data as U
    r, q, d, n = args.refs, args.queries, 256, 25
    ref, qry = U.retrieval_sets(r, q, d)
    per = (r + world - 1) // world
    lo = min(rank * per, r)
    shard = torch.tensor(ref[lo:lo + per], device=dev)
    qt = torch.tensor(qry, device=dev)
    del ref

    def fence():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            out = fn()
        fence()
        el = time.perf_counter() - t0
        if dp:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el, out

    stats = {}

    def step():
        return parallel.topn_l2_sharded(shard, qt, n, lo, score=args.score, force_exchange=dp)

    def local_only():
        return retrieval.topn_l2(shard, qt, n, idx_offset=lo, score=args.score, stats=stats)

    for _ in range(args.warmup):
        step()
    elapsed, (dd, ii) = timed(step, args.steps)
    el_local, _ = timed(local_only, args.steps)
    if rank == 0:
        out = {
            'metric': 'retrieval queries/sec (exact L2 top-%d, %d references x %d queries x %d)' % (n, r, q, d),
            'value': round(q * args.steps / elapsed, 1), 'unit': 'queries/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32' if args.score == 'f32' else 'bf16x3 nomination, f64 re-rank',
            'data': 'synthetic',
            'config': {'workload': 'configs[4]: exact top-%d, reference set sharded over %d rank(s) '
                                   '(%d rows each), queries replicated, [Q,n] candidates '
                                   'all-gathered and merged' % (n, world, per),
                       'refs': r, 'queries': q, 'd': d, 'n': n, 'score': args.score,
                       'parallelism': 'ref-shard%d' % world},
            'local_scan_ms_per_step': round(el_local / args.steps * 1e3, 3),
            'exchange_and_merge_us_per_step': round((elapsed - el_local) / args.steps * 1e6, 1),
            'uncertified_queries_local': stats.get('uncertified'),
            'world_seen': dist.get_world_size() if dp else 1,
            'backend': dist.get_backend() if dp else None,
            'checksum_idx': int(ii.sum()),
            'switches': _switches(args),
        }
        if args.n1_ref > 0:
            out['scaling_efficiency'] = round(out['value'] / (world * args.n1_ref), 4)
        print(json.dumps(out))
    if dp:
        dist.destroy_process_group()


def loss_b192_line(dev, iters=10):
    """configs[3]'s loss shape on one GPU, event-timed per kernel: wms forward + backward at
    B = 192 x 32768.  Two cases: 'all_rows' (a single process owning all 192 rows: every row of
    d loss / d E) and 'own_rows' — what EVERY RANK of an 8-GPU run evaluates after the all-gather:
    the full 192 x 192 forward, the backward for its own 24 rows only (parallel.wms_loss_dp)."""
    from soft_contrastive_learning_amd import _lib
    from soft_contrastive_learning_amd.model import losses
    from tests import util_data as U
    bsz, own = 192, 24
    emb = torch.tensor(U.embeddings(bsz, E), device=dev, requires_grad=True)
    dm = torch.tensor(U.positions_distances(bsz)[None], device=dev)
    out = {'B': bsz, 'E': E, 'note': 'event-bracketed, bracket subtracted (kernel_timing.bracket_us); '
                                    'rocprofv3 trace under profiles/'}
    for case, rows in (('all_rows', None), ('own_rows', (72, own))):
        for _ in range(3):
            losses.wms_loss(dm, emb, 0.8, 15.0, _rows=rows).backward()
        torch.cuda.synchronize()
        with _lib.KernelTimer(capacity=16 * iters) as kt:
            for _ in range(iters):
                losses.wms_loss(dm, emb, 0.8, 15.0, _rows=rows).backward()
            torch.cuda.synchronize()
        models = kernel_models(bsz if rows is None else own, 1200, bsz, 4)
        krows = [price(k, cnt, ms, models[k]) for k, (cnt, ms) in sorted(kt.summary().items())
                 if k in models]
        out[case] = {'kernels': krows, 'us_forward_backward': round(sum(r['us'] for r in krows), 1)}
    return out


def batch_sweep(dev, iters=5):
    """the kernels of the head over the batch, kernel-only, a few launches
    each, right after the timed region.  NetVLAD (bf16 map, 1200 locations, plane images from the
    packing launch) forward / backward at b = 24 / 48 / 96 images per GPU: launches, corrected
    microseconds, fraction of the stage's governing bound (netvlad_stage).  wms loss at B = 24 /
    48 / 96 / 192 x 32768: forward and backward against max(HBM, float32-MFMA) of SURVEY 8(d)."""
    from soft_contrastive_learning_amd import _lib
    from soft_contrastive_learning_amd.model import losses, nets
    from tests import util_data as U
    out = {'netvlad': [], 'wms_loss': [],
           'how': 'event-bracketed launches minus the bracket (kernel_timing.bracket_us), %d iterations '
                  'per size; bound = max(algorithmic bytes / 8 TB/s, executed flops / MFMA peak)' % iters}
    w, c = U.vlad_params()
    wt = torch.tensor(w, device=dev).reshape(1, 1, D, K).requires_grad_(True)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, D, K).requires_grad_(True)
    wd = wt.detach()
    gen = torch.Generator(device=dev).manual_seed(5)
    for b in (24, 48, 96):
        x = torch.randn(b, 1, 1200, D, device=dev, generator=gen).bfloat16().requires_grad_(True)
        g = torch.randn(b, E, device=dev, generator=gen)
        nets.prepack([], force=True, vlad_w=wd)
        pl = nets.fresh_vlad_planes(wd)        # (the weights do not change inside this loop)
        for _ in range(2):
            nets.netvlad(x, wt, ct, True, pl).backward(g)
        torch.cuda.synchronize()
        with _lib.KernelTimer(capacity=16 * iters) as kt:
            for _ in range(iters):
                nets.netvlad(x, wt, ct, True, pl).backward(g)
            torch.cuda.synchronize()
        steps = (1200 + 31) // 32
        per = -(-steps * b // 256)
        models = kernel_models(b, 1200, b, 2, slices=-(-steps // per))
        rows = [price(k, cnt, ms, models[k]) for k, (cnt, ms) in sorted(kt.summary().items()) if k in models]
        st = netvlad_stage(rows, b, 1200, 2, iters)
        ent = {'images': b}
        for ps in ('forward', 'backward'):
            if ps in st:
                ent[ps] = {k: st[ps][k] for k in ('launches_per_step', 'us_per_step', 'bound', 'bound_us', 'frac',
                                                  'us_per_step_events', 'frac_events')}
                ent[ps]['us_per_image'] = round(st[ps]['us_per_step'] / b, 3)
        out['netvlad'].append(ent)
        del x, g
    # inference (evaluation/inference.py at large images_per_pass): nothing saved; from 144 images on
    # one workgroup per image with the finish in its tail (one launch)
    out['netvlad_inference'] = []
    for b in (96, 256):
        x = torch.randn(b, 1, 1200, D, device=dev, generator=gen).bfloat16()
        nets.prepack([], force=True, vlad_w=wd)
        pl = nets.fresh_vlad_planes(wd)
        with torch.no_grad():
            for _ in range(2):
                nets.netvlad(x, wd, ct.detach(), True, pl)
            torch.cuda.synchronize()
            with _lib.KernelTimer(capacity=16 * iters) as kt:
                for _ in range(iters):
                    nets.netvlad(x, wd, ct.detach(), True, pl)
                torch.cuda.synchronize()
        steps = (1200 + 31) // 32
        per = -(-steps * b // 256)
        models = kernel_models(b, 1200, b, 2, slices=-(-steps // per))
        rows = [price(k, cnt, ms, models[k]) for k, (cnt, ms) in sorted(kt.summary().items()) if k in models]
        us = sum(r['us'] * r['launches'] for r in rows) / iters
        bound = (b * 1200 * D * 2 + b * D * K * 4 + 2 * D * K * 4) / (PEAK_HBM_GBPS * 1e9) * 1e6
        out['netvlad_inference'].append({'images': b, 'kernels': [r['kernel'] for r in rows],
                                         'us': round(us, 2), 'bound': 'hbm', 'bound_us': round(bound, 2),
                                         'frac': round(bound / us, 4) if us > 0 else None,
                                         'us_per_image': round(us / b, 3)})
        del x
    for bsz in (24, 48, 96, 192):
        emb = torch.tensor(U.embeddings(bsz, E), device=dev, requires_grad=True)
        dm = torch.tensor(U.positions_distances(bsz)[None], device=dev)
        for _ in range(2):
            losses.wms_loss(dm, emb, 0.8, 15.0).backward()
        torch.cuda.synchronize()
        with _lib.KernelTimer(capacity=16 * iters) as kt:
            for _ in range(iters):
                losses.wms_loss(dm, emb, 0.8, 15.0).backward()
            torch.cuda.synchronize()
        models = kernel_models(bsz, 1200, bsz, 4)
        rows = [price(k, cnt, ms, models[k]) for k, (cnt, ms) in sorted(kt.summary().items()) if k in models]
        fwd = [r for r in rows if not r['kernel'].startswith('gram_bwd')]
        bwd = [r for r in rows if r['kernel'].startswith('gram_bwd')]
        bound = max(bsz * E * 4 / (PEAK_HBM_GBPS * 1e9), 2.0 * bsz * bsz * E / (PEAK_F32_TFLOPS * 1e12)) * 1e6
        ent = {'B': bsz, 'bound_us_each_pass': round(bound, 2)}
        for name, rs in (('forward', fwd), ('backward', bwd)):
            us = sum(r['us'] * r['launches'] for r in rs) / iters
            us_ev = sum(r.get('us_events', r['us']) * r['launches'] for r in rs) / iters
            ent[name] = {'launches': round(sum(r['launches'] for r in rs) / iters, 2), 'us': round(us, 2),
                         'frac': round(bound / us, 4) if us > 0 else None,
                         'us_events': round(us_ev, 2),
                         'frac_events': round(bound / us_ev, 4) if us_ev > 0 else None,
                         'kernels': [r['kernel'] for r in rs]}
        out['wms_loss'].append(ent)
    return out


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)                       # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.stub_cpu:
        return stub_main(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU path exists)')
    # TEST ONLY (SCL_BENCH_ONE_GPU_GLOO=1): every rank on cuda:0 with gloo carrying the
    # collectives — the whole data-parallel step on a one-GPU box; the timing means nothing
    one_gpu = os.environ.get('SCL_BENCH_ONE_GPU_GLOO') == '1'
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.dtype == 'f32':      # the bf16 step runs no library convolution
        torch.backends.cudnn.benchmark = bool(args.miopen_find)
    dev = torch.device('cuda', local_rank)
    from soft_contrastive_learning_amd import _lib, parallel
    if args.force_dist and world != 1:
        raise SystemExit('--force-dist is the one-rank form of the collective path: use --gpus 1')
    # dp: the step runs through the process group (world > 1, or the one-rank group of --force-dist)
    dp = world > 1 or args.force_dist
    if dp:
        parallel.init_process_group(dev, backend='gloo' if one_gpu else 'nccl',
                                    timeout_s=args.dist_timeout, force_single=args.force_dist)
    try:
        return _main_body(args, world, rank, dev, dp)
    except BaseException as exc:
        if dp and not (isinstance(exc, SystemExit) and exc.code in (0, None)):
            parallel.abort_rank(1)      # never leave the peers waiting in a collective
        raise


def _main_body(args, world, rank, dev, dp):
    from soft_contrastive_learning_amd import _lib, parallel
    from soft_contrastive_learning_amd.model import losses, nets
    if args.variant:
        _lib.use_diag()               # the variants exist in the diagnostic build only
    _lib.load()
    if args.workload == 'retrieval':
        return retrieval_main(args, world, rank, dev, dp)
    if args.side_wrw >= 0:
        nets.USE_SIDE_WRW = bool(args.side_wrw) and nets.USE_SIDE_WRW
    nets.USE_SPLIT_FWD = bool(args.split_fwd)
    if args.variant:
        _lib.load().scl_debug_set_variant(args.variant)
        nets.USE_PREPACK = False      # a pinned kernel may not read the packed-image layout

    b, gb = args.batch, args.batch * world
    cdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = nets.VGG16NetVLAD(compute_dtype=cdt, seed=1234, fused_relu=bool(args.fused_relu)).to(dev)
    params = list(model.parameters())
    buckets = parallel.GradBuckets(params, force_collectives=args.force_dist)
    # train/train.py:1270 base_lr; one fused kernel, step counter on the device (graph-safe)
    # tf.train.AdamOptimizer (train/train.py:870): torch's fused kernel with TF's epsilon placement
    # (train/optim.py; under --graph the captured step keeps the eps of the capture)
    from soft_contrastive_learning_amd.train.optim import TFAdam
    opt = TFAdam(params, lr=5e-6, fused=True, capturable=bool(args.graph))

    # synthetic RobotCar-shaped batch, resident in HBM
    g = torch.Generator().manual_seed(42 + rank)
    images = torch.randint(0, 256, (b, args.height, args.width, 3), generator=g).float().to(dev)
    rng = np.random.default_rng(7)
    xy = rng.uniform(0.0, 200.0, size=(gb, 2))
    dmat = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)
    distances = torch.tensor(dmat[None], device=dev)

    if os.environ.get('SCL_GRAD_SINK', '1') != '0':
        nets.GRAD_SINK = buckets   # conv weight / bias gradients go straight into the flat buffer

    def step():
        buckets.zero()
        emb = model(images)
        if dp:
            loss = parallel.wms_loss_dp(distances, emb, 0.8, 15.0)
        else:
            loss = losses.wms_loss(distances, emb, d_alpha=0.8, d_beta=15.0)
        loss.backward()
        buckets.finish()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    reserve_choice = None
    if dp and 'SCL_RESERVE_CUS' not in os.environ:
        # untimed, part of the warm-up: the persistent convolution grids take every CU, RCCL's
        # kernels need some — leave 0 or 8 CUs free, whichever gives the
        # shorter step on THIS node (max over ranks, so every rank decides the same)
        lib_ = _lib.load()
        for _ in range(24):                  # (at steady state, like the second-stream choice below)
            step()
        fence()
        tried = {0: float('inf'), 8: float('inf')}
        for _ in range(2):
            for rv in (0, 8):
                lib_.scl_set_reserve_cus(rv)
                step()
                fence()
                t1 = time.perf_counter()
                for _ in range(5):
                    step()
                fence()
                tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tried[rv] = min(tried[rv], float(tt) / 5 * 1e3)
        best_rv = min(tried, key=tried.get)
        lib_.scl_set_reserve_cus(best_rv)
        reserve_choice = {'chosen': best_rv, 'ms_per_step_tried': {str(k): round(v, 3) for k, v in tried.items()}}
    side_choice = None
    if args.side_wrw < 0 and nets.USE_SIDE_WRW and nets.GRAD_SINK is not None:
        # untimed, part of the warm-up: which setting of the second stream this device prefers.
        # Decided in the regime the timed steps run in: a chip that has just started is cooler and
        # clocks higher than after a quarter of a second under load (the first collections of round 5
        # chose on 3-step bursts from cold: 11.38 ms in the warm-up, 11.60 in the timed region), so
        # the device is brought to its steady state first and every setting gets 3 x 5 steps.
        if reserve_choice is None:          # (else the device is at its steady state already)
            for _ in range(24):
                step()
            fence()
        side_choice = nets.autotune_side_wrw(step, steps=5, rounds=3)
        side_choice['settle_steps'] = 24
        if dp:                              # every rank must run the same schedule
            flag = torch.tensor([1.0 if side_choice['chosen'] else 0.0], device=dev)
            dist.all_reduce(flag)
            nets.USE_SIDE_WRW = bool(flag.item() * 2 >= world)
            side_choice['chosen'] = nets.USE_SIDE_WRW
            if nets._FFW_ENV == 'auto':      # (and the fused first-layer gradients, which go with it)
                flag = torch.tensor([1.0 if side_choice.get('fused_first_wrw') else 0.0], device=dev)
                dist.all_reduce(flag)
                nets.USE_FUSED_FIRST_WRW = nets.USE_SIDE_WRW and bool(flag.item() * 2 >= world)
                side_choice['fused_first_wrw'] = nets.USE_FUSED_FIRST_WRW
        fence()
    # One step captured in a HIP graph and replayed: the step is ~125 dependent launches, and
    # the launch gaps between them cost ~0.3 ms of a 14 ms step when issued one by one.  The
    # captured work is exactly step(): nothing is skipped, cached or reused between replays
    # except the allocations.  Falls back to eager launches if the capture is refused.
    graph = None
    run = step
    if args.graph and not dp and os.environ.get('SCL_BENCH_EVENTS_IN_TIMED_REGION') != '1':
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                   # allocator warm-up on the side stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = step()
            graph.replay()
            torch.cuda.synchronize()

            def run():
                graph.replay()
                return static_loss
        except Exception as exc:                         # noqa: BLE001 (report, then go eager)
            sys.stderr.write('bench.py: HIP graph capture failed (%s: %s); eager launches\n'
                             % (type(exc).__name__, exc))
            graph = None
            run = step
            torch.cuda.synchronize()
    # The K timed steps run uninstrumented: bracketing every kernel launch with two HIP events
    # costs about 6 % of a step (measured: 15.05 vs 14.24 ms), which is instrumentation, not
    # the path.  Per-kernel durations for `roofline` / `kernels` come from PROF extra steps of
    # the same process right after the timed region, with the event sink on
    # (SCL_BENCH_EVENTS_IN_TIMED_REGION=1 puts the sink around the timed steps instead).
    in_region = os.environ.get('SCL_BENCH_EVENTS_IN_TIMED_REGION') == '1'
    prof_steps = args.steps if in_region else max(1, min(args.steps, 5))
    import contextlib
    if in_region:
        nets.WORK_LOG = {}
    with (_lib.KernelTimer(capacity=256 * max(args.steps, 1)) if in_region
          else contextlib.nullcontext()) as kt:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for k in range(args.steps):
            loss = run()
            evs[k + 1].record()
        fence()
        elapsed = time.perf_counter() - t0
    per_step = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps))
    if not in_region:
        nets.WORK_LOG = {}    # algorithmic flops / bytes of the backbone kernels, per call site
        # one stream here: the timed steps run the weight-gradient kernels on a second stream
        # next to the backward-data kernels (nets.USE_SIDE_WRW); a duration taken while two
        # kernels share the CUs belongs to neither, so the instrumented steps serialise them
        side_wrw, nets.USE_SIDE_WRW = nets.USE_SIDE_WRW, False
        split_fwd, nets.USE_SPLIT_FWD = nets.USE_SPLIT_FWD, False
        try:
            with _lib.KernelTimer(capacity=256 * prof_steps) as kt:
                t1 = time.perf_counter()
                for _ in range(prof_steps):
                    step()
                fence()
                elapsed_prof = time.perf_counter() - t1
        finally:
            nets.USE_SIDE_WRW = side_wrw
            nets.USE_SPLIT_FWD = split_fwd
    else:
        elapsed_prof = elapsed
    work, nets.WORK_LOG = nets.WORK_LOG, None
    loss_val = float(loss.detach())
    global BRACKET_US
    BRACKET_US = bracket_us(dev) if rank == 0 else 0.0

    def timed_steps(k):
        fence()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        el = time.perf_counter() - t1
        if dp:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        return el / k * 1e3

    # (not under rocprofv3: its preloaded library initialises the GPU in every child process, and
    # rocm-smi's `#!/usr/bin/env python3` hop is then an exec after GPU initialisation, which the
    # GPU boxes of this pool refuse)
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(
        k.startswith(('ROCPROF', 'ROCPROFILER')) for k in os.environ)
    telemetry = (gpu_telemetry(step, fence)
                 if rank == 0 and world == 1 and not args.no_telemetry and not profiled else None)
    comm = None
    if dp:
        # what the first multi-GPU run must be able to explain by itself
        diag_steps = max(3, min(args.steps, 10))
        parallel.COMM_LOG = {'allgather': [], 'finish': []}
        ms_diag = timed_steps(diag_steps)
        log, parallel.COMM_LOG = parallel.COMM_LOG, None

        def med(pairs):
            v = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
            return round(v[len(v) // 2], 1) if v else None
        lib = _lib.load()
        old_reserve = lib.scl_set_reserve_cus(0)
        reserve = {}
        for rv in (0, 8):
            lib.scl_set_reserve_cus(rv)
            timed_steps(2)
            reserve[str(rv)] = round(timed_steps(diag_steps), 3)
        lib.scl_set_reserve_cus(old_reserve)
        comm = {'world_seen': dist.get_world_size(), 'backend': dist.get_backend(),
                'diag_steps': diag_steps, 'ms_per_step_with_comm_events': round(ms_diag, 3),
                'allgather_us_median': med(log['allgather']),
                'allgather_bytes_per_rank': b * E * 4,
                'finish_wait_us_median': med(log['finish']),
                'allreduce_buckets': len(buckets.buckets),
                'allreduce_bytes': int(buckets.flat.numel() * 4),
                'ms_per_step_by_reserved_cus': reserve,
                'reserved_cus_in_timed_region': old_reserve,
                'reserved_cus_warmup_choice': reserve_choice,
                'how': 'device events on the compute stream around all_gather_into_tensor and '
                       'around the waits of GradBuckets.finish() (rank 0, median over the diagnostic '
                       'steps run after the timed region); step time with scl_set_reserve_cus(0 / 8): '
                       'same process, max over ranks'}
    if dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        n_loc = (args.height // 16) * (args.width // 16)
        models = kernel_models(b, n_loc, gb, 2 if cdt == torch.bfloat16 else 4)
        summ = kt.summary()
        kernels = [price(k, c, ms, models[k]) for k, (c, ms) in sorted(summ.items()) if k in models]
        # the dominant kernel of the NetVLAD + pairwise-loss family ...
        dom_head = max(kernels, key=lambda r: r['us'] * r['launches']) if kernels else None
        # the backbone's own kernels (convolutions against the bf16 dense MFMA peak or HBM,
        # whichever governs; elementwise glue against HBM), work as logged by the call sites
        for k, (c, ms) in sorted(summ.items()):
            if k in work and work[k][0] == c:
                _, fl, by = work[k]
                kernels.append(price(k, c, ms, dict(flops=fl / c, bytes=by / c,
                                                    peak_tflops=PEAK_BF16_TFLOPS)))
        # ... and of the whole step (a backbone convolution): `roofline` is the latter
        dom = max(kernels, key=lambda r: r['us'] * r['launches']) if kernels else None

        def roof(d):
            peak_tf = PEAK_BF16_TFLOPS if d['kernel'] in work else PEAK_F32_TFLOPS
            return dict(kernel=d['kernel'], bound=d['bound'],
                        achieved=d.get('executed_tflops', d['tflops'])
                        if d['bound'] == 'mfma' else d['gbps'],
                        peak=d.get('mfma_peak_tflops', peak_tf)
                        if d['bound'] == 'mfma' else PEAK_HBM_GBPS,
                        unit='TFLOP/s' if d['bound'] == 'mfma' else 'GB/s',
                        frac=d['frac'], traffic=None, us_per_launch=d['us'],
                        launches_per_step=d['launches'] / max(prof_steps, 1))
        roofline = roof(dom) if dom else None
        roofline_head = roof(dom_head) if dom_head else None
        if roofline:
            # HBM bytes per launch measured offline with rocprofv3 PMC on this exact shape
            # (profiles/pmc_traffic.json); null when the shape / dtype was not profiled
            try:
                with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
                    pmc = json.load(f)
                for rf in (roofline, roofline_head):
                    ent = pmc.get(args.dtype, {}).get(rf['kernel'])
                    if ent and pmc['shape'] == {'batch': b, 'locations': n_loc}:
                        rf['traffic'] = ent['read'] + ent['write']
                        alg = (work[rf['kernel']][2] / work[rf['kernel']][0]
                               if rf['kernel'] in work else models[rf['kernel']]['bytes'])
                        rf['traffic_algorithmic'] = int(alg)
                        rf['traffic_source'] = ('rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch, '
                                                'L2 misses incl. Infinity-Cache hits '
                                                '(profiles/pmc_traffic.json: %s)' % pmc.get('source', '?'))
            except (OSError, ValueError, KeyError):
                pass
        if world == 1 and roofline and roofline['bound'] == 'mfma' and roofline['kernel'] in work:
            try:
                sus = sustained_mfma(dev)
                top = max(sus['tflops_32x32x16'], sus['tflops_16x16x32'])
                sus['frac_of_sustained'] = round(roofline['achieved'] / top, 4)
                sus['sustained_over_peak'] = round(top / roofline['peak'], 4)
                roofline['sustained'] = sus
            except (RuntimeError, AttributeError) as e:
                roofline['sustained'] = {'error': str(e)}
        hip_ms = sum(r['us'] * r['launches'] for r in kernels) / 1e3 / max(prof_steps, 1)
        out = {
            'metric': 'train-step images/sec (VGG16-NetVLAD soft-MS, 640x480)',
            'value': round(gb * args.steps / elapsed, 2),
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'ms_per_step_stats': {'median': round(per_step[len(per_step) // 2], 3),
                                  'p10': round(per_step[int(0.1 * (len(per_step) - 1))], 3),
                                  'p90': round(per_step[int(0.9 * (len(per_step) - 1) + 0.5)], 3),
                                  'min': round(per_step[0], 3), 'max': round(per_step[-1], 3),
                                  'how': 'device events at the step boundaries of the timed '
                                         'region (no host synchronisation inside it)'},
            'hip_graph': graph is not None,
            'debug_variant': args.variant,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: wms soft-contrastive train step, VGG16-NetVLAD K=64, '
                                   '%d images/GPU %dx%d, %s backbone, float32-accurate NetVLAD + f32 loss'
                                   % (b, args.width, args.height, args.dtype),
                       'global_batch': gb, 'locations': n_loc, 'parallelism': 'dp%d' % world,
                       'optimizer': 'adam', 'loss': float('%.6g' % loss_val)},
            'kernel_timing': {'method': 'HIP events around every launch of the hand-written kernels '
                                        '(scl_prof sink), on the launch stream',
                              'steps': prof_steps,
                              'where': 'timed region' if in_region else
                              'extra eager steps right after the timed region (the events cost '
                              '~6 % of a step, so the timed steps run without them)',
                              'streams': 'one (the timed steps run the weight-gradient kernels on a '
                                         'second stream next to the backward-data kernels; the '
                                         'instrumented steps serialise them so that a duration '
                                         'is one kernel alone)'
                              if (nets.USE_SIDE_WRW or bool(nets.USE_SPLIT_FWD)) else 'one',
                              'ms_per_step_with_events': round(elapsed_prof / prof_steps * 1e3, 3),
                              'bracket_us': round(BRACKET_US, 2),
                              'null_kernel_us': round(NULL_KERNEL_DEVICE_US or 0.0, 2),
                              'bracket_how': 'event duration of the empty kernel (scl_prof_null) minus '
                                             'its own time (null_kernel_us: 200 launches back to back '
                                             'between one event pair), both in this process; subtracted '
                                             'from every `us` below, at most %d %% of a duration; '
                                             '`us_events` / `frac_events` are the raw figures'
                                             % int(BRACKET_CAP * 100)},
            'roofline': roofline,
            'roofline_netvlad_loss': roofline_head,
            'roofline_netvlad_stage': netvlad_stage(kernels, b, n_loc, 2 if cdt == torch.bfloat16 else 4,
                                                    prof_steps),
            'kernels': kernels,
            'hip_path_ms_per_step': round(hip_ms, 3),
        }
        out['switches'] = _switches(args)
        out['side_stream'] = dict(mode='auto' if args.side_wrw < 0 else 'fixed',
                                  on=bool(nets.USE_SIDE_WRW), **(
                                      {k: v for k, v in (side_choice or {}).items() if k != 'chosen'}))
        if comm is not None:
            out['comm'] = comm
        if telemetry is not None:
            out['telemetry'] = telemetry
        if args.n1_ref > 0:
            out['scaling_efficiency'] = round(out['value'] / (world * args.n1_ref), 4)
        if world == 1 and not args.no_batch_sweep:
            out['batch_sweep'] = batch_sweep(dev)
        if world == 1:
            out['roofline_loss_b192'] = loss_b192_line(dev)
            if not args.no_retrieval:
                out['retrieval'] = retrieval_line(dev)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, torch.get_num_threads())
            out['cpu_baseline_kernels'] = cpu_baseline_kernels(torch.get_num_threads())
        print(json.dumps(out))
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
