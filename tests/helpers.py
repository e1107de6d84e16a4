"""Shared helpers for the test-suite: golden fixture access and oracle dispatch."""

import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta']))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != 'meta'}
    return meta, arrays


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def synth_state_dict(meta):
    from resselt_amd.utils import synth

    kw = dict(meta['synth'])
    for k in ('blocks', 'depth', 'depths', 'num_heads', 'split_size', 'scale_list'):
        if k in kw:
            kw[k] = tuple(kw[k])
    fn = {'esrgan': synth.rrdbnet_state_dict, 'spanplus': synth.spanplus_state_dict, 'span': synth.span_state_dict,
          'swinir': getattr(synth, 'swinir_state_dict', None), 'compact': synth.compact_state_dict, 'dat': getattr(synth, 'dat_state_dict', None), 'spanpp': getattr(synth, 'spanpp_state_dict', None), 'hat': getattr(synth, 'hat_state_dict', None), 'rtmosr': getattr(synth, 'rtmosr_state_dict', None), 'drct': getattr(synth, 'drct_state_dict', None)}[meta['arch']]  # fmt: skip
    return fn(seed=meta['seed'], **kw)


def oracle_forward(meta, sd, x):
    if meta['arch'] == 'esrgan':
        from oracle.rrdbnet import rrdbnet_forward

        return rrdbnet_forward(sd, x)
    if meta['arch'] == 'spanplus':
        from oracle.span import spanplus_forward

        return spanplus_forward(sd, x)
    if meta['arch'] == 'span':
        from oracle.span import span_forward

        return span_forward(sd, x)
    if meta['arch'] == 'swinir':
        from oracle.swinir import swinir_forward

        return swinir_forward(sd, x)
    if meta['arch'] == 'compact':
        from oracle.compact import compact_forward

        return compact_forward(sd, x)
    if meta['arch'] == 'spanpp':
        from oracle.spanpp import spanpp_forward

        return spanpp_forward(sd, x, meta.get('scale'))
    if meta['arch'] == 'rtmosr':
        from oracle.rtmosr import rtmosr_forward

        return rtmosr_forward(sd, x)
    if meta['arch'] == 'hat':
        from oracle.hat import hat_forward

        return hat_forward(sd, x)
    if meta['arch'] == 'drct':
        from oracle.drct import drct_forward

        return drct_forward(sd, x)
    if meta['arch'] == 'dat':
        from oracle.dat import dat_forward

        return dat_forward(sd, x)
    raise KeyError(meta['arch'])
