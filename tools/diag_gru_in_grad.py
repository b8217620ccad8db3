"""Where does the H=300 gradient error of the default (2-piece) mode come from?  d(loss)/d(GRU input) of the generator
(train mode, dropout on, B=88) against the oracle, per product mode and per column block of the GRU input."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle import s2ag_oracle as O
from s2ag_testing import build_product, oracle_cfg, to_cuda
from test_gpu_modules import _g_noise
from speech2affective_gestures_amd import noise, ops
lib = ops._lib()
hidden, n_words, n_spk, B, s0 = 300, 2000, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 88, 5000
inp = O.recipe_inputs(B, 34, s0 + 10, n_words, n_spk)
gi = to_cuda(inp)
pre_seq = O.make_pre_seq(inp['target'], 4)
gen = torch.Generator().manual_seed(1)
oc = oracle_cfg(hidden, 0.3)
ref = None
for pieces in (0, 3, 2):
    lib.s2ag_gru_coop_set_split_pieces(pieces)
    noise.reset_sites(0)
    cfg, mods, sds = build_product(hidden, n_words, n_spk, 0.3, s0, which=('G',))
    G = mods['G'].train()
    G.cut_backward = True
    noise.manual_seed(77)
    nz = torch.tensor([77, 0], dtype=torch.int64, device='cuda')
    out, z, mu, lv = G(pre_seq.cuda(), gi['in_text'], gi['in_mfcc'], gi['vid'])
    fulls, leaves = G._cut
    if ref is None:
        pin = _g_noise(G, nz, B, 34, hidden, 0.3, 0.1)
        sd = {k: (v.clone().requires_grad_(True) if O.is_param(k) and '.net.' not in k else v.clone()) for k, v in sds['G'].items()}
        nzo = O.Noise(pin)
        audio = O.mfcc_encoder(sd, 'audio_encoder.', inp['in_mfcc'], True)
        text = O.text_encoder_tcn(sd, 'text_encoder.', inp['in_text'], True, oc.dropout_prob, nzo)
        zz, mu_r, lv_r = O._speaker_z(sd, inp['vid'], nzo)
        pre = O.aff_encoder(sd, 'aff_encoder.', pre_seq[..., :-1], True)
        in_data = torch.cat((pre, audio, text, zz.unsqueeze(1).expand(-1, 34, -1)), dim=2)
        in_data.retain_grad()
        o_r = O._decode(sd, oc.dropout_prob, in_data, True, nzo, 0.01, False)
        d_out = torch.randn(o_r.shape, generator=gen)
        (o_r * d_out).sum().backward()
        ref = (in_data.detach(), in_data.grad.clone(), o_r.detach())
    (out * d_out.cuda()).sum().backward()
    gl = leaves[0].grad.cpu().double()
    r = ref[1].double()
    e = lambda a, b: float((a - b).abs().max() / b.abs().max())
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    blocks = dict(pre=slice(0, 8), audio=slice(8, 40), text=slice(40, 72), z=slice(72, 88))
    print(f'pieces={pieces}: out {e(out.detach().cpu().double(), ref[2].double()):.2e}; gru_in fwd {e(fulls[0].detach().cpu().double(), ref[0].double()):.2e}; '
          f'd(gru_in) max {e(gl, r):.2e} L2 {l2(gl, r):.2e}; ' +
          ', '.join(f'{k}: {e(gl[..., s], r[..., s]):.2e}/{l2(gl[..., s], r[..., s]):.2e}' for k, s in blocks.items()), flush=True)
lib.s2ag_gru_coop_set_split_pieces(-1)
