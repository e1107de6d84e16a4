#!/usr/bin/env python3
"""Headline benchmark: output megapixels/s of RealESRGAN-x4plus (RRDBNet, 23 blocks) on 1080p input.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line.
A *step* is one full forward pass of the hot path over one synthetic 3x1080x1920 frame per GPU
(1920x1080 -> 7680x4320 = 33.18 output MP), input resident in HBM before the timed region, output left in HBM.

N > 1, one rank per GPU.  Under ``torch.distributed.run`` (WORLD_SIZE set) this process IS a rank; otherwise the process is
the launcher: before it touches the GPU it starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
child, relays rank 0's JSON line and exits with the child's status.

  --config c2 (default, ``"scaling": "weak"``): tile-parallel inference (resselt_amd/tiling.py) of ONE image made of N 1080p
      tiles (grid 1x2, 2x2, 2x4 = the 16:9 tile aspect): every rank upsamples its own tile plus a 32-pixel input halo and the
      upscaled tiles are reassembled on every rank by an RCCL all-gather over xGMI, inside the timed region.
  --config c5 (``"scaling": "strong"``): BASELINE.json configs[4] -- one 3x4320x7680 input cut into 2x4 tiles of 2160x1920
      (+halo 32) dealt round-robin to the N ranks (N divides 8); a rank runs each of its tiles in 1080p-sized sub-tiles
      so that its activation buffers stay those of the 1080p plan.
``value`` = output MP of the whole image / max-over-ranks time, in both modes.

Extra objects on the JSON line:
  roofline      dominant kernel (the fused conv kernel family = every launch of the forward), MFMA-bound:
                algorithmic FLOP (2,240,856 per output pixel) / HIP-event time of the launches
  roofline_hbm  the same time priced against the layer-wise HBM model of SURVEY.md §8d (7,790 B / output px)
  cpu_baseline  the CPU oracle (oracle/rrdbnet.py, a restatement of the reference forward; the reference
                cannot travel to the GPU box) timed on the host cores on a bounded 256x256 crop
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_OUT_PX = 2_240_856  # RRDBNet-23 x4, 2*MAC per output pixel
HBM_B_PER_OUT_PX = 7_790  # layer-wise bf16 byte model
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--blocks', type=int, default=23)
    ap.add_argument('--precision', default='auto', choices=['auto', 'mixed', 'bf16x3', 'bf16'],
                    help="auto = the engine's default policy (RRDBNet: residual dense blocks in one fp16 product, head / tail in three bf16 products)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-crop', type=int, default=256)
    ap.add_argument('--halo', type=int, default=32)
    ap.add_argument('--no-power', action='store_true', help='skip the socket-power leg (rocm-smi sampled by a child process started before the GPU is touched)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the compact c3 / c4 legs that the default one-GPU run appends under "secondary"')
    ap.add_argument('--no-kernel-roofline', action='store_true', help='skip the per-kernel replays (profiling runs: keeps the launch mix that of plain forwards)')
    ap.add_argument('--dry-run', action='store_true', help='launcher / rendezvous check without a GPU: ranks meet over gloo, rank 0 prints a stub line')
    ap.add_argument('--io', default='f32', choices=['f32', 'u8'], help='tensors crossing the boundary: fp32 [N,C,H,W] in / out, or uint8 [N,H,W,C] images in / out '
                    '(the 8-bit path: /255 and clamp*255+round inside the first / last kernel; tiles cross xGMI as bytes)')
    ap.add_argument('--config', default='c2', choices=['c2', 'c3', 'c4', 'c5'],
                    help='c2 (default, BASELINE metric): N tiles of 1080p through RealESRGAN-x4plus (weak scaling); c5: one 4320x7680 input, 2x4 tiles (strong '
                    'scaling); c3: SPANPlus x4 fp16, 8x3x512x512 tiles, one GPU; c4: SwinIR-L x4 bf16 tensors, 1x3x1024x1024, one GPU')
    return ap.parse_args()


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without torch.distributed.run: start the N ranks as a child job.  Nothing in this process has
    initialised the GPU (importing torch does not), and no exec happens: the parent waits and relays rank 0's JSON line."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]  # fmt: skip
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f'bench.py: the {n}-rank child job failed (status {proc.returncode})', file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def cpu_baseline(sd, crop: int) -> dict:
    """Time the CPU oracle on a bounded crop of the same workload (reported baseline, not a target): at 8 threads (the survey's
    setting) and at the host's physical core count; ``value`` is the faster of the two, both are listed."""
    from oracle.rrdbnet import rrdbnet_forward
    from resselt_amd.utils import synth

    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 8
    except Exception:
        phys = os.cpu_count() or 8
    x = synth.synth_input((1, 3, crop, crop), seed=0)
    runs = {}
    default_threads = torch.get_num_threads()
    for threads in sorted({8, phys}):
        torch.set_num_threads(threads)
        best = None
        with torch.no_grad():
            for _ in range(2):
                t0 = time.perf_counter()
                y = rrdbnet_forward(sd, x)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        runs[threads] = (y.shape[-1] * y.shape[-2] / 1e6 / best, best)
    torch.set_num_threads(default_threads)
    cores = max(runs, key=lambda k: runs[k][0])
    return {
        'value': round(runs[cores][0], 4),
        'unit': 'output megapixels/s',
        'cores': cores,
        'kind': 'port',
        'sample': f'oracle/rrdbnet.py fp32 on one 3x{crop}x{crop} crop of the synthetic frame (best of 2 per thread count), torch {torch.__version__} CPU',
        'by_threads': {str(k): {'value': round(v[0], 4), 'seconds': round(v[1], 2)} for k, v in runs.items()},
        'host_physical_cores': phys,
    }


def _kernel_key(name: str):
    """(shape, plane format, products) of a ring-kernel name in either spelling (bench.py's classes / tools/hbm_traffic.py's families)."""
    import re

    if 'conv_ring_pair' in name:
        return ('pair',)
    if 'conv_ring_up2' in name:
        return ('up2',)
    m = re.search(r'conv_ring<(\d)', name)
    if not m:
        return (name.split('(')[0].strip(),)
    head = name.split('>')[0]
    fmt = 'f16' if re.search(r'(?<!b)f16', head) else 'bf16'
    pm = re.search(r'(?:f16|bf16),(\d)', head)
    return (m.group(1), fmt, pm.group(1) if pm else '3')


def measured_traffic(kernel: str = '') -> dict:
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (tools/hbm_traffic.py: separate FETCH_SIZE /
    WRITE_SIZE passes of this same command, gfx950 x2 FETCH correction).  Counters cannot be read from inside the timed process, so the
    summary carries the source hash of the library it was collected on; when that differs from the library loaded now the number is
    stale and ``traffic`` is reported as null.  ``kernel``: the class bench.py found dominant; its family is looked up in the summary."""
    import glob

    from resselt_amd import build as B

    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')))
    if not files:
        return {'bytes': None, 'note': 'no PMC summary under profiles/'}
    rec = json.load(open(files[-1]))
    name = os.path.basename(files[-1])
    have = None
    if os.path.exists(B.STAMP):
        with open(B.STAMP) as f:
            have = f.read().strip()
    if rec.get('srchash') is None or rec.get('srchash') != have:
        return {'bytes': None, 'note': f'{name} was collected on another build (srchash {str(rec.get("srchash"))[:12]} != loaded {str(have)[:12]}): stale, not reported'}
    dom = rec.get('dominant', {})
    for k in rec.get('kernels', []):
        if kernel and _kernel_key(k.get('kernel', '')) == _kernel_key(kernel):
            dom = k
            break
    return {'bytes': dom.get('traffic_bytes_per_launch'), 'note': f'{name}: {dom.get("kernel")}, FETCH_SIZE x2 + WRITE_SIZE per launch; whole forward '
            f'{rec.get("traffic_GB_per_forward", 0):.1f} GB = {rec.get("traffic_over_algorithmic", 0):.2f}x the 258.5 GB layer-wise bf16 model'}  # fmt: skip


def kernel_classes(model, reps: int, x=None) -> list:
    """Per-kernel roofline, measured live and IN THE FRAME: the launch list of the model's plan is replayed in order through the C-ABI, one
    launch (or one fused pair of launches: what rsa_conv2d_list makes of two fusable neighbours) at a time with a HIP event between
    consecutive launches on the launch stream, ``reps`` passes; a launch's duration is the time between its two events, so every launch runs
    in the cache state its predecessor leaves (round 3 replayed each distinct shape alone, back to back with itself, which over-counted
    the frame by 6 %).  Grouped by the kernel a launch dispatches to; the classes sum to the replayed frame."""
    from resselt_amd.engine import lib as L

    from resselt_amd.engine.base import conv_algorithmic_bytes

    if x is not None:
        keep = model(x)  # fills the plan's buffers; holding the result keeps the last layer's output pointer valid during the replays
        torch.cuda.synchronize()
    plan = model.last_plan()
    stream = torch.cuda.current_stream().cuda_stream
    # (callable, kernel name, flop, algorithmic bytes, products); name None = host-side steps and unpriced layout / copy kernels ("other")
    launches = []
    for step in plan.steps:
        conv = getattr(step, '_rsa_conv', None)
        if conv is None:
            meta = getattr(step, '_rsa_meta', None)
            if meta is None:
                launches.append((step, None, 0.0, 0.0, 1))
            else:
                launches.append((step, meta['kernel'], float(meta['flop']), float(meta['bytes']), meta.get('products', 1)))
            continue
        arr, cins = conv
        i = 0
        while i < len(arr):
            p = arr[i]
            flop = 2.0 * p.ksize * p.ksize * cins[i] * p.cout * p.H * p.W * p.batch  # the layer's TRUE input channels (a 3-channel first layer occupies a plane of 8)
            if i + 1 < len(arr) and L.conv_pair_fusable(arr[i], arr[i + 1]):
                q = arr[i + 1]
                flop += 2.0 * 9 * cins[i + 1] * q.cout * q.H * q.W * q.batch
                # ALGORITHMIC bytes of the launch = SURVEY.md 8d's layer-wise model of the TWO layers it computes (every operand of every layer once:
                # conv1 + conv2 = 192 + 256 B, conv3 + conv4 = 320 + 384 B per pixel at 2 B per element) -- the figure the unfused launches were
                # priced with.  What the fused launch has to move at least (common input once, both outputs) is `fused_bytes_per_launch`.
                nbytes = conv_algorithmic_bytes(p) + conv_algorithmic_bytes(q)
                fused = p.cin_planes * 16 * p.batch * p.H * p.W + 2 * ((p.cout + 7) // 8) * 16 * p.batch * p.H * p.W
                two = (L.ConvParams * 2)(arr[i], arr[i + 1])
                launches.append((lambda two=two: L.conv2d_list(two, stream), 'rsa::conv_ring_pair (two growth convolutions of a dense block in one launch, one fp16 product)', flop, nbytes, 1, fused))
                i += 2
                continue
            one = (L.ConvParams * 1)(p)
            launches.append((lambda one=one: L.conv2d_list(one, stream), L.conv_kernel_name(p), flop, conv_algorithmic_bytes(p), p.products))
            i += 1
    total_us = [0.0] * len(launches)
    for rep in range(reps + 1):  # the first pass warms up
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(launches) + 1)]
        evs[0].record()
        for k, (fn, *_rest) in enumerate(launches):
            fn()
            evs[k + 1].record()
        torch.cuda.synchronize()
        if rep:
            for k in range(len(launches)):
                total_us[k] += evs[k].elapsed_time(evs[k + 1]) * 1e3 / reps
    other_us = sum(us for (fn, name, *_r), us in zip(launches, total_us) if name is None)
    timed = [(l, us) for l, us in zip(launches, total_us) if l[1] is not None]
    launches, total_us = [l for l, _ in timed], [us for _, us in timed]
    groups: dict = {}
    for (one, name, flop, nbytes, prod, *extra), us in zip(launches, total_us):
        g = groups.setdefault(name, {'kernel': name, 'launches': 0, 'us': 0.0, 'flop': 0.0, 'bytes': 0.0, 'products': prod, 'fused': 0.0})
        g['launches'] += 1
        g['us'] += us
        g['flop'] += flop
        g['bytes'] += nbytes
        g['fused'] += extra[0] if extra else 0.0
    out = []
    for g in groups.values():
        out.append({
            'kernel': g['kernel'],
            'launches': g['launches'],
            'avg_us': round(g['us'] / g['launches'], 2),
            'ms_per_forward': round(g['us'] / 1e3, 3),
            'flop_per_launch': round(g['flop'] / g['launches']),
            'tflops': round(g['flop'] / g['us'] / 1e6, 2),
            'products': g['products'],
            'bytes_per_launch': round(g['bytes'] / g['launches']),  # every operand of every layer of the launch once, in the layouts it reads and writes
            'gbs': round(g['bytes'] / g['us'] / 1e3, 1),
            **({'fused_bytes_per_launch': round(g['fused'] / g['launches'])} if g['fused'] else {}),
        })  # fmt: skip
    out.sort(key=lambda c: -c['ms_per_forward'])
    if other_us > 0.0:
        out.append({'kernel': 'other (layout conversion, band copies, host-side steps between launches)', 'launches': 0, 'avg_us': 0.0, 'ms_per_forward': round(other_us / 1e3, 3),
                    'flop_per_launch': 0, 'tflops': 0.0, 'products': 1, 'bytes_per_launch': 0, 'gbs': 0.0})  # fmt: skip
    return out


def dry_run(args, world: int, rank: int) -> None:
    """No GPU work: proves that the ranks start, meet (gloo), agree on a max-over-ranks time and that rank 0's line reaches the caller."""
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == float(world)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'output megapixels/sec, RealESRGAN-x4plus 1080p\u21924K, 1/2/4/8 MI355X', 'value': None, 'n_gpus': world, 'dry_run': True,
                          'config': {'workload': args.config}}), flush=True)  # fmt: skip


_SAMPLER = r"""
import os, re, subprocess, sys, time
out = open(sys.argv[1], 'w', buffering=1)
start, dev = sys.argv[2], sys.argv[3]
while not os.path.exists(start):  # idle (no process spawns, no SMU queries) until the bench reaches its power leg
    time.sleep(0.05)
while True:
    try:
        txt = subprocess.run(['rocm-smi', '-d', dev, '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        break
    t = time.time()
    pw = [float(v) for v in re.findall(r'Current Socket Graphics Package Power \(W\): ([\d.]+)', txt)]
    ck = [float(v) for v in re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)', txt)]
    cap = [float(v) for v in re.findall(r'Max Graphics Package Power \(W\): ([\d.]+)', txt)]
    if pw:
        out.write(f'{t} {pw[0]} {ck[0] if ck else 0} {cap[0] if cap else 0}\n')
"""


class PowerSampler:
    """Socket power / shader clock of THIS process's GPU (the first visible device), sampled by a CHILD process (python + rocm-smi, ~5 samples
    per second) that is started before this process touches the GPU, never touches it itself, and stays idle until ``begin()`` -- so
    nothing is spawned or queried during model load, warm-up or the timed region; the samples are matched to the power leg by wall clock."""

    def __init__(self):
        import shutil
        import subprocess
        import tempfile

        self.proc, self.path = None, None
        if shutil.which('rocm-smi') is None:
            return
        fd, self.path = tempfile.mkstemp(prefix='rsa_power_', suffix='.txt')
        os.close(fd)
        self.start_path = self.path + '.start'
        vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES') or '0'
        dev = vis.split(',')[0].strip()
        self.proc = subprocess.Popen([sys.executable, '-c', _SAMPLER, self.path, self.start_path, dev if dev.isdigit() else '0'],
                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)  # fmt: skip
        import atexit

        atexit.register(lambda p=self.proc: p.poll() is None and p.terminate())  # never outlives the bench

    def begin(self):
        if self.proc is not None:
            open(self.start_path, 'w').close()  # a file: this process has initialised the GPU by now and must not start programs

    def stop(self, t0: float, t1: float):
        if self.proc is None:
            return None
        self.proc.terminate()
        try:
            self.proc.wait(timeout=15)
        except Exception:
            self.proc.kill()
        rows = []
        try:
            for line in open(self.path):
                v = line.split()
                if len(v) == 4 and t0 <= float(v[0]) <= t1:
                    rows.append([float(a) for a in v])
            os.unlink(self.path)
            os.unlink(self.start_path)
        except OSError:
            return None
        if not rows:
            return None
        pw, ck = [r[1] for r in rows], [r[2] for r in rows]
        return {'avg_w': round(sum(pw) / len(pw), 1), 'max_w': max(pw), 'cap_w': rows[-1][3] or None, 'sclk_mhz_avg': round(sum(ck) / len(ck)),
                'samples': len(rows), 'note': 'rocm-smi socket power of this GPU over the last 2 s of a 3.5 s run of back-to-back forwards (sensor averaging window ~1 s)'}  # fmt: skip



SECONDARY = {
    # BASELINE.json configs[2] / configs[3]; constants: SURVEY.md 8d (algorithmic FLOP and layer-wise bytes per OUTPUT pixel)
    'c3': dict(metric='output megapixels/sec, SPANPlus x4 fp16, batch of 8 3x512x512 tiles, 1 MI355X', shape=(8, 3, 512, 512), io=torch.float16,
               flop_px=53_154, bytes_px=276, workload='SPANPlus x4 (fc48, blocks [4], pixel-shuffle head), 8x3x512x512 fp16 tiles -> 8x3x2048x2048'),
    'c4': dict(metric='output megapixels/sec, SwinIR-L x4 bf16, 3x1024x1024, 1 MI355X', shape=(1, 3, 1024, 1024), io=torch.bfloat16,
               flop_px=3_833_694, bytes_px=27_600, workload='SwinIR-L x4 (embed 240, 9x6 blocks, 8 heads, window 8, nearest+conv, 3conv), 1x3x1024x1024 bf16 tensors -> 1x3x4096x4096'),
}  # fmt: skip


def secondary(args):
    """``--config c3 | c4``: one JSON line for BASELINE configs[2] / configs[3]."""
    if args.gpus != 1:
        raise SystemExit(f'--config {args.config} is a one-GPU configuration')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the engine has no CPU path')
    print(json.dumps(secondary_result(args.config, args, steps=args.steps, warmup=args.warmup)), flush=True)


def secondary_result(config: str, args, steps: int, warmup: int, compact: bool = False) -> dict:
    """BASELINE configs[2] (c3) and configs[3] (c4) on ONE GPU, same JSON schema as the headline: roofline of the dominant kernel measured
    live (the plan replayed launch by launch between HIP events), the layer-wise HBM fraction, and the CPU oracle beside it.
    ``compact``: the object the DEFAULT run carries under ``secondary`` (value, ms_per_step, dtype, dominant-kernel roofline, and the
    max-abs difference against the CPU oracle on a crop of the same synthetic input)."""
    import resselt_amd
    from resselt_amd.engine import lib as L
    from resselt_amd.utils import synth

    cfg = SECONDARY[config]
    dev = torch.device('cuda', torch.cuda.current_device())
    if config == 'c3':
        sd = synth.spanplus_state_dict(upscale=4, upsampler='ps', seed=0)
    else:
        sd = synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv', seed=0)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    model.precision = args.precision
    prec = model.resolved_precision()
    x = synth.synth_input(cfg['shape'], seed=0).to(dev).to(cfg['io'])
    for _ in range(max(1, warmup)):
        model(x)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        y = model(x)
    ev1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    L.check_status('bench: timed region')
    aborts = L.ring_aborts()
    if aborts:
        raise SystemExit(f'bench.py: {aborts} ring-schedule hand-offs timed out; the timed forwards are invalid')
    out_px = y.shape[0] * y.shape[2] * y.shape[3]
    kern_s = ev0.elapsed_time(ev1) / 1e3 / steps
    del y
    classes = [] if args.no_kernel_roofline else kernel_classes(model, max(2, min(steps, 5)), x)
    flop, hbm = cfg['flop_px'] * out_px, cfg['bytes_px'] * out_px
    res = {
        'metric': cfg['metric'], 'value': round(out_px / 1e6 / dt, 3), 'unit': 'output megapixels/s', 'n_gpus': 1, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(dt * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'mixed': 'fp16', 'fp16': 'fp16'}.get(prec, 'bf16'), 'data': 'synthetic',
        'config': {'workload': cfg['workload'] + ', synthetic uniform(+-1/sqrt(fan_in)) weights', 'precision': args.precision if args.precision == prec else f'{args.precision} -> {prec}',
                   'launches_per_step': model.launches_per_forward()},
        'roofline_kernels': classes,
        'roofline_frame': {'bound': 'mfma', 'achieved': round(flop / kern_s / 1e12, 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': round(flop / kern_s / 1e12 / MFMA_PEAK_TFLOPS, 4), 'note': f'{cfg["flop_px"]:,} algorithmic FLOP per output pixel (SURVEY.md 8d) / event time of a forward'},
        'roofline_hbm': {'bound': 'hbm', 'achieved': round(hbm / kern_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(hbm / kern_s / 1e9 / HBM_PEAK_GBS, 4),
                         'model': f'layer-wise bytes, {cfg["bytes_px"]:,} B per output pixel (SURVEY.md 8d)'},
        'ring_aborts': aborts, 'event_ms_per_step': round(kern_s * 1e3, 3),
    }  # fmt: skip
    if classes:
        dom = classes[0]
        fm, fh = dom['tflops'] / MFMA_PEAK_TFLOPS, dom['gbs'] / HBM_PEAK_GBS
        hb = fh > fm
        res['roofline'] = {'bound': 'hbm' if hb else 'mfma', 'kernel': dom['kernel'], 'achieved': dom['gbs'] if hb else dom['tflops'],
                           'peak': HBM_PEAK_GBS if hb else MFMA_PEAK_TFLOPS, 'unit': 'GB/s' if hb else 'TFLOP/s', 'frac': round(max(fm, fh), 4),
                           'frac_mfma': round(fm, 4), 'frac_hbm': round(fh, 4), 'traffic': None, 'avg_launch_us': dom['avg_us'],
                           'launches_per_forward': dom['launches'], 'flop_per_launch': dom['flop_per_launch'], 'bytes_per_launch': dom['bytes_per_launch'],
                           'share_of_frame': round(dom['ms_per_forward'] / (kern_s * 1e3), 3)}  # fmt: skip
    from oracle.span import spanplus_forward
    from oracle.swinir import swinir_forward

    fwd = spanplus_forward if config == 'c3' else swinir_forward
    if compact:
        # parity beside the number: the same model on a crop of the same synthetic input against the CPU oracle (fp32 tensors: the
        # arithmetic; and tensors of the configuration's dtype: that plus the output rounding of the 16-bit tensor)
        cshape = (1, 3, 128, 128) if config == 'c3' else (1, 3, 64, 64)
        xc = synth.synth_input(cfg['shape'], seed=0)[:1, :, : cshape[2], : cshape[3]].contiguous()
        default_threads = torch.get_num_threads()
        torch.set_num_threads(8)
        with torch.no_grad():
            ref32 = fwd(sd, xc)
            refio = fwd(sd, xc.to(cfg['io']).float())
        torch.set_num_threads(default_threads)
        e32 = (model(xc.to(dev)).float().cpu() - ref32).abs().max().item()
        eio = (model(xc.to(dev).to(cfg['io'])).float().cpu() - refio).abs().max().item()
        torch.cuda.synchronize()
        L.check_status('bench: parity leg')
        out = {k: res[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'event_ms_per_step', 'steps', 'warmup', 'dtype')}
        out['config'] = res['config']
        out['roofline'] = {k: res['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_mfma', 'frac_hbm', 'avg_launch_us',
                                                           'launches_per_forward', 'share_of_frame')} if 'roofline' in res else None  # fmt: skip
        out['roofline_frame_frac_mfma'] = res['roofline_frame']['frac']
        out['roofline_hbm_frac'] = res['roofline_hbm']['frac']
        out['parity'] = {'max_abs_vs_oracle_fp32_tensors': float(f'{e32:.3e}'), f'max_abs_vs_oracle_{str(cfg["io"]).split(".")[-1]}_tensors': float(f'{eio:.3e}'),
                         'sample': f'{"x".join(map(str, cshape))} crop of the synthetic input, oracle ({fwd.__module__}) fp32 on the CPU'}  # fmt: skip
        model._invalidate()
        return out
    if not args.no_cpu_baseline:
        cshape = (2, 3, 512, 512) if config == 'c3' else (1, 3, 128, 128)
        xc = synth.synth_input(cshape, seed=0)
        default_threads = torch.get_num_threads()
        torch.set_num_threads(8)
        best = None
        with torch.no_grad():
            for _ in range(2):
                t = time.perf_counter()
                yc = fwd(sd, xc)
                d = time.perf_counter() - t
                best = d if best is None else min(best, d)
        torch.set_num_threads(default_threads)
        res['cpu_baseline'] = {'value': round(yc.shape[0] * yc.shape[2] * yc.shape[3] / 1e6 / best, 4), 'unit': 'output megapixels/s', 'cores': 8, 'kind': 'port',
                               'sample': f'oracle ({fwd.__module__}) fp32 on a {"x".join(map(str, cshape))} sample of the workload, best of 2, {best:.2f} s, torch {torch.__version__} CPU'}  # fmt: skip
    return res


def main():
    args = parse()
    if args.config in SECONDARY:
        return secondary(args)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher')
    if args.config == 'c5' and 8 % world:
        raise SystemExit('--config c5 has 8 tiles: --gpus must divide 8')
    if args.dry_run:
        return dry_run(args, world, rank)
    # The power sampler is a child process started before the first GPU call.  Under a profiler the GPU is already initialised by the
    # preloaded tool library when this line runs, and a process that has touched the GPU must not start programs: no power leg then.
    profiled = any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTRACER', 'RPD_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')
    sampler = PowerSampler() if (world == 1 and not args.no_power and not profiled) else None
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the engine has no CPU path')
    # RSA_DIST_BACKEND=gloo rehearses the N > 1 code path with several ranks on ONE GPU (RCCL refuses two ranks per device)
    backend = os.environ.get('RSA_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import resselt_amd
    from resselt_amd.utils import synth

    sd = synth.rrdbnet_state_dict(nb=args.blocks, seed=0)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    model.precision = args.precision
    H, W = args.height, args.width
    from resselt_amd.tiling import TileParallel, choose_grid, upscale_tiled

    if args.config == 'c5':
        # BASELINE.json configs[4]: one 4320x7680 input, 2x4 tiles of 2160x1920 dealt round-robin to the ranks; each rank runs a tile
        # (2160x1920 + halo) as 1080p-sized sub-tiles, so its buffers are those of the 1080p plan (strong scaling: total work is fixed)
        rows, cols = 2, 4
        img_h, img_w = 4 * H, 4 * W
        assert (img_h, img_w) == (rows * 2 * H, cols * W)
        if args.io == 'u8':
            from resselt_amd.tiling import upscale

            inner = lambda crop: upscale(model, crop, tile=(H + 2 * args.halo, W + 2 * args.halo), halo=args.halo, scale=4)  # noqa: E731
        else:
            inner = lambda crop: upscale_tiled(model, crop, 4, tile=(H + 2 * args.halo, W + 2 * args.halo), halo=args.halo)  # noqa: E731
        runner = TileParallel(inner, scale=4, halo=args.halo, grid=(rows, cols))
    else:
        # N tiles of HxW (weak scaling) stacked as N x 1 full-width row bands: an interior band reads 2 x halo extra rows (5.9 % of a 1080p
        # tile; a 2 x 4 grid of the same tiles reads halos on three sides, 9.5 %), its 1144 x 1920 window is 72 x 60 = 4320 ring tiles =
        # 16.9 per CU, and the bands of an 8-bit image are contiguous slabs of the result, so the all-gather lands in place
        rows, cols = (world, 1)
        img_h, img_w = rows * H, cols * W
        runner = TileParallel(model, scale=4, halo=args.halo, grid=(rows, cols)) if world > 1 else model
    # the same full input image on every rank (3 channels: cheap to replicate); each rank owns its tile(s) of it
    x = synth.synth_input((1, 3, img_h, img_w), seed=0).to(dev)
    if args.io == 'u8':
        x = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()  # [1, H, W, 3] image

    def step():
        return runner(x)  # N > 1: tile forward + RCCL all-gather; every rank ends with the whole upscaled image

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    y = None
    for _ in range(args.steps):
        y = step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the ring schedule's failure word (rsa_check_status; the stream is synchronised): a timed-out hand-off means wrong pixels, so no line
    from resselt_amd.engine import lib as L

    L.check_status('bench: timed region')
    aborts = L.ring_aborts()
    if aborts != 0:
        raise SystemExit(f'bench.py: {aborts} ring-schedule hand-offs timed out; the timed forwards are invalid')

    total_out_px = (y.shape[1] * y.shape[2] if args.io == 'u8' else y.shape[-1] * y.shape[-2]) * y.shape[0]  # whole image (on every rank after the all-gather)
    assert total_out_px == 16 * img_h * img_w
    ms_per_step = dt / args.steps * 1e3
    value = total_out_px / 1e6 / (dt / args.steps)
    del y

    if rank == 0:
        # kernel-only time of the conv launches of one forward of ONE HxW frame, from HIP events on the launch stream
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        xt = (x[:, :H, :W] if args.io == 'u8' else x[:, :, :H, :W]).contiguous()
        model(xt)
        torch.cuda.synchronize()
        k0.record()
        for _ in range(args.steps):
            model(xt)
        k1.record()
        torch.cuda.synchronize()
        kern_s = k0.elapsed_time(k1) / 1e3 / args.steps
        out_px = 16 * H * W
        n_launch = model.launches_per_forward()
        macs = model.macs_per_input_pixel()
        if args.blocks == 23:
            assert 2 * macs == FLOP_PER_OUT_PX * 16, macs  # cross-check with SURVEY.md 8d
        flop = 2 * macs * H * W
        achieved_tf = flop / kern_s / 1e12
        achieved_gbs = HBM_B_PER_OUT_PX * out_px / kern_s / 1e9
        layout_bytes = model.conv_bytes_per_forward()  # same layer-wise model, priced in this engine's split-plane/f32-map layouts
        prec = model.resolved_precision()
        prec_note = {
            'mixed': 'residual-dense-block convolutions (92 % of the MACs) in ONE fp16 product on hi planes, conv_first / upsampling / HR / last '
                     'convolutions in three bf16 products on split planes, trunk convolution in three fp16 products; f32 accumulate',
            'bf16x3': 'bf16 MFMA operands split hi+lo (3 products), f32 accumulate',
            'bf16': 'plain bf16 MFMA operands, f32 accumulate',
        }[prec]  # fmt: skip
        if args.no_kernel_roofline:
            classes = [{'kernel': 'all conv launches of one forward (per-kernel replays skipped)', 'launches': n_launch, 'avg_us': round(kern_s / n_launch * 1e6, 2),
                        'ms_per_forward': round(kern_s * 1e3, 3), 'flop_per_launch': round(flop / n_launch), 'tflops': round(achieved_tf, 2)}]  # fmt: skip
        else:
            classes = kernel_classes(model, max(2, min(args.steps, 5)), xt)
        priced = [c for c in classes if c['launches']]
        dom = max(priced, key=lambda c: c['ms_per_forward'])
        issued_tf = sum(c['tflops'] * c.get('products', 1) * c['ms_per_forward'] for c in priced) / max(1e-9, sum(c['ms_per_forward'] for c in priced))
        traffic = measured_traffic(dom['kernel']) if (prec == 'mixed' and (H, W, args.blocks) == (1080, 1920, 23)) else {'bytes': None, 'note': 'PMC traffic is only collected for the default workload'}
        # which roof the dominant kernel is nearer to: its algorithmic FLOP rate over the dense MFMA peak, or its algorithmic bytes (every
        # operand once, in the layouts it reads and writes) per second over the HBM peak
        dom_mfma, dom_hbm = dom['tflops'] / MFMA_PEAK_TFLOPS, dom.get('gbs', 0.0) / HBM_PEAK_GBS
        dom_bound = 'hbm' if dom_hbm > dom_mfma else 'mfma'
        tile_note = {
            'c2': (f'{rows}x{cols} full-width row bands of {H}x{W} (+{args.halo} rows of input halo), one per GPU; one asynchronous RCCL all-gather of '
                   + ('uint8 HWC output bands straight into the result image' if args.io == 'u8' else 'fp32 output tiles')) if world > 1 else 'single tile',
            'c5': f'one 3x{img_h}x{img_w} input, 2x4 tiles of {2 * H}x{W} (+{args.halo} px halo) dealt round-robin to {world} rank(s), each run as 1080p-sized sub-tiles; '
            + ((f'one asynchronous RCCL all-gather of {"uint8" if args.io == "u8" else "fp32"} output tiles per round of {world} tiles, beside the next round\'s compute') if world > 1 else 'no collective'),
        }[args.config]
        res = {
            'metric': 'output megapixels/sec, RealESRGAN-x4plus 1080p\u21924K, 1/2/4/8 MI355X',
            'value': round(value, 3),
            'unit': 'output megapixels/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True,
            'scaling': 'strong' if args.config == 'c5' else 'weak',
            'vs_baseline': None,
            'dtype': 'fp16' if prec == 'mixed' else 'bf16',  # the arithmetic type of the dominant layers (config.precision names the whole policy)
            'data': 'synthetic',
            'config': {
                'workload': (f'RealESRGAN-x4plus (RRDBNet nf64 nb{args.blocks} gc32 x4), 1x3x{H}x{W} {"uint8" if args.io == "u8" else "fp32"} frame per GPU -> 1x3x{4 * H}x{4 * W}, ' if args.config == 'c2' else
                             f'RealESRGAN-x4plus (RRDBNet nf64 nb{args.blocks} gc32 x4), ONE 1x3x{img_h}x{img_w} {"uint8" if args.io == "u8" else "fp32"} image -> 1x3x{4 * img_h}x{4 * img_w} over all GPUs (BASELINE configs[4]), ')
                + prec_note + ', synthetic uniform(+-1/sqrt(fan_in)) weights',
                'precision': args.precision if args.precision == prec else f'{args.precision} -> {prec}',
                'precision_policy': prec_note,
                'io': 'uint8 [N,H,W,C] images in and out' if args.io == 'u8' else 'fp32 [N,C,H,W] tensors in and out',
                'tile_parallel': tile_note,
                'launches_per_step': n_launch,
            },
            # the DOMINANT kernel (largest share of the frame), timed live per launch with HIP events on the launch stream
            'roofline': {
                'bound': dom_bound,
                'kernel': dom['kernel'],
                'achieved': dom['gbs'] if dom_bound == 'hbm' else dom['tflops'],
                'peak': HBM_PEAK_GBS if dom_bound == 'hbm' else MFMA_PEAK_TFLOPS,
                'unit': 'GB/s' if dom_bound == 'hbm' else 'TFLOP/s',
                'frac': round(max(dom_hbm, dom_mfma), 4),
                'frac_mfma': round(dom_mfma, 4),
                'frac_hbm': round(dom_hbm, 4),
                'bytes_per_launch': dom.get('bytes_per_launch'),
                'fused_bytes_per_launch': dom.get('fused_bytes_per_launch'),  # a fused launch: what it must move at least (common input once)
                'traffic': traffic['bytes'],
                'traffic_note': traffic['note'],
                'avg_launch_us': dom['avg_us'],
                'launches_per_forward': dom['launches'],
                'flop_per_launch': dom['flop_per_launch'],
                'mfma_issued_frac': round(dom['tflops'] * dom.get('products', 1) / MFMA_PEAK_TFLOPS, 4),
                'share_of_frame': round(dom['ms_per_forward'] / (kern_s * 1e3), 3),
            },
            'roofline_kernels': classes,  # every kernel class of the forward, same definitions
            'roofline_frame': {
                'bound': 'mfma',
                'achieved': round(achieved_tf, 2),
                'peak': MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': round(achieved_tf / MFMA_PEAK_TFLOPS, 4),
                'mfma_issued_frac': round(issued_tf / MFMA_PEAK_TFLOPS, 4),
                'note': f'all {n_launch} launches of one forward: 2,240,856 algorithmic FLOP per output pixel / event time of the forward',
            },
            'roofline_hbm': {
                'bound': 'hbm',
                'achieved': round(achieved_gbs, 1),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved_gbs / HBM_PEAK_GBS, 4),
                'model': 'layer-wise bf16 bytes, 7,790 B per output pixel (SURVEY.md 8d)',
                'layout_bytes_per_launch': None if not layout_bytes else round(layout_bytes / n_launch),
                'layout_achieved': None if not layout_bytes else round(layout_bytes / kern_s / 1e9, 1),
                'layout_note': 'the same every-operand-once model priced in the engine layouts (split planes hi+lo = 4 B/channel where a layer reads or writes both halves, 2 B/channel for hi-only fp16 planes)',
            },
            'ring_aborts': aborts,
            'event_ms_per_step': round(ev_ms / args.steps, 3),
            'forward_ms_1080p': round(kern_s * 1e3, 3),
        }
        if sampler is not None:  # power leg: 3.5 s of back-to-back forwards, the last 2 s of samples
            sampler.begin()
            w0 = time.time()
            while time.time() - w0 < 3.5:
                model(xt)
                torch.cuda.synchronize()
            w1 = time.time()
            res['power'] = sampler.stop(w1 - 2.0, w1)
        res['roofline_kernels_note'] = ('the launch list of one forward replayed in order, one launch at a time, a HIP event between consecutive launches: '
                                        f'sum of ms_per_forward = {sum(c["ms_per_forward"] for c in classes):.3f} ms (the replayed frame; forward_ms_1080p is the same list in one host call)')
        if world == 1 and args.config == 'c2' and not args.no_secondary and (H, W, args.blocks) == (1080, 1920, 23):
            # BASELINE configs[2] and [3] in the same process, after the headline (its buffers are released first): compact objects
            model._invalidate()
            torch.cuda.empty_cache()
            res['secondary'] = {}
            for name in ('c3', 'c4'):
                t_sec = time.perf_counter()
                res['secondary'][name] = secondary_result(name, args, steps=max(3, min(args.steps, 10)), warmup=3, compact=True)
                res['secondary'][name]['leg_seconds'] = round(time.perf_counter() - t_sec, 1)
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(sd, args.cpu_crop)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
