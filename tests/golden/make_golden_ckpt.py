"""Golden fixture for the checkpoint key maps, made by the reference's own converter functions
(tools/multimae2vit_converter.py, tools/vit2multimae_converter.py) and utils/pos_embed.py::interpolate_pos_embed_multimae,
run on a tiny random state dict.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py"""
import importlib.util
import os
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    m2v = load(os.path.join(REF, 'tools', 'multimae2vit_converter.py'), 'ref_m2v')
    v2m = load(os.path.join(REF, 'tools', 'vit2multimae_converter.py'), 'ref_v2m')
    pe = load(os.path.join(REF, 'utils', 'pos_embed.py'), 'ref_pos_embed')
    torch.manual_seed(21)
    D = 8
    sd = {'global_tokens': torch.randn(1, 1, D), 'input_adapters.rgb.pos_emb': torch.randn(1, D, 3, 3),
          'input_adapters.rgb.proj.weight': torch.randn(D, 3, 2, 2), 'input_adapters.rgb.proj.bias': torch.randn(D),
          'input_adapters.depth.proj.bias': torch.randn(D), 'encoder.0.attn.qkv.weight': torch.randn(3 * D, D),
          'encoder.1.mlp.fc2.bias': torch.randn(D), 'output_adapters.rgb.mask_token': torch.randn(1, 1, D)}
    out = {}
    for k, v in sd.items():
        out['in/' + k] = v.numpy().copy()
    for k, v in m2v.multimae_to_vit({k: v.clone() for k, v in sd.items()}).items():
        out['vit/' + k] = v.numpy()
    for k, v in m2v.multimae_to_vitmultimae({k: v.clone() for k, v in sd.items()}).items():
        out['vitmm/' + k] = v.numpy()
    vit = {k: v.clone() for k, v in m2v.multimae_to_vit({k: v.clone() for k, v in sd.items()}).items()}
    vit['pos_embed'] = torch.randn_like(vit['pos_embed'])
    for k, v in vit.items():
        out['vit_in/' + k] = v.numpy().copy()
    for k, v in v2m.vit_to_multimae({k: v.clone() for k, v in vit.items()}).items():
        out['back/' + k] = v.contiguous().numpy()
    # pos-emb resize 3x3 -> 5x4 for the rgb adapter of a stand-in model
    model = types.SimpleNamespace(input_adapters=types.SimpleNamespace(rgb=types.SimpleNamespace(pos_emb=torch.zeros(1, D, 5, 4))))
    ck = {k: v.clone() for k, v in sd.items()}
    pe.interpolate_pos_embed_multimae(model, ck)
    out['resized/input_adapters.rgb.pos_emb'] = ck['input_adapters.rgb.pos_emb'].numpy()
    np.savez_compressed(os.path.join(HERE, 'ckpt_maps.npz'), **out)
    print('wrote ckpt_maps.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
