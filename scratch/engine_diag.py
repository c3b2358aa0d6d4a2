import sys, numpy as np, torch
sys.path.insert(0,'licv-vqa_amd'); sys.path.insert(0,'.')
from licv.config import IDEFICS_TINY, IDEFICS_MID
from licv.synthetic import synth_idefics_weights
from licv.idefics_engine import IdeficsWeights, IdeficsEngine
T=torch.from_numpy
for tag, arch in (("g3_idefics_tiny", IDEFICS_TINY), ("g3_idefics_mid", IDEFICS_MID)):
    z = np.load(f'tests/golden/{tag}.npz')
    sd = synth_idefics_weights(arch, seed=int(z['meta'][0]), dtype=torch.float32)
    w = IdeficsWeights(sd, arch, 'cuda')
    for fuse in (False, True):
        eng = IdeficsEngine(w, fuse_hook_norm=fuse)
        ins = dict(input_ids=T(z['in_input_ids']).cuda(), attention_mask=T(z['in_attention_mask']).cuda(),
                   pixel_values=T(z['in_pixel_values']).cuda(), image_attention_mask=T(z['in_image_attention_mask']).cuda())
        cap = {}
        lo = eng.forward(**ins, capture=cap)
        gl = T(z['bf16_logits_off']); g32 = T(z['f32_logits_off'])
        print(tag, 'fuse', fuse, 'OFF  hip-vs-bf16gold', float((lo.float().cpu()-gl).abs().max()), 'hip-vs-f32gold', float((lo.float().cpu()-g32).abs().max()),
              'bf16gold-vs-f32gold', float((gl-g32).abs().max()), 'scale', float(g32.abs().max()))
        print('   img states', float((cap['image_states'].float().cpu().reshape(-1)-T(z['bf16_image_states']).reshape(-1)).abs().max()),
              float((T(z['bf16_image_states']).reshape(-1)-T(z['f32_image_states']).reshape(-1)).abs().max()), float(T(z['f32_image_states']).abs().max()))
        for hs in ('all','sub'):
            if f'bf16_{hs}_logits' not in z.files: continue
            layers = list(range(arch.num_layers)) if hs=='all' else [1,3]
            icv = T(z['icv_full'])[:, :len(layers)].cuda()
            cap = {}
            lg = eng.forward(**ins, icv=icv, hook_layers=layers, capture=cap)
            gl = T(z[f'bf16_{hs}_logits']); g32 = T(z[f'f32_{hs}_logits'])
            raw = torch.stack([t.float().cpu() for t in cap['raw']]); graw = T(z[f'bf16_{hs}_raw']); graw32=T(z[f'f32_{hs}_raw'])
            print('  ', hs, 'logits hip-bf16gold', float((lg.float().cpu()-gl).abs().max()), 'hip-f32gold', float((lg.float().cpu()-g32).abs().max()), 'gold-gold', float((gl-g32).abs().max()),
                  '| raw hip-bf16gold', float((raw-graw).abs().max()), 'hip-f32', float((raw-graw32).abs().max()), 'gold-gold', float((graw-graw32).abs().max()), 'scale', float(graw32.abs().max()))
