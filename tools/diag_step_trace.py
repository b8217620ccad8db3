"""Diagnosis (TEST INFRASTRUCTURE, runs on the CPU device model): a reference step trace (tests/golden/step_small*.npz variant
given as argv[1]: speaker | znone | noreg | warmup) followed by the product and by the oracle side by side -- per step the
losses, the worst gradient differences (a) against the oracle on ITS trajectory and (b) against the oracle started from the
PRODUCT'S OWN weights, and the weight drift.  (b) staying at ~1e-5 while (a) grows says: the product's gradients are right,
the trajectories part because Adam turns rounding-level gradient elements into +-lr steps (tests/test_gpu_step.py _final_close).
    python tools/diag_step_trace.py znone        EXTRA="{'early_main': 0}" selects a schedule of the trainer"""
import os, sys
os.environ['S2AG_EMU']='1'
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/emu')
import harness; harness.install()
import numpy as np, torch
from oracle import s2ag_oracle as O
from s2ag_testing import *
import test_gpu_step as T
from speech2affective_gestures_amd import noise, processor_v2 as P
variant=sys.argv[1]
overrides,fname=T.STEP_VARIANTS[variant]
g=dict(np.load('/root/repo/tests/golden/'+fname))
hidden,n_words,n_spk,B,s0=32,64,12,4,4000
pr,_=T.make_processor(hidden,n_words,n_spk,B,s0,0.0,cfg_overrides=overrides, **eval(os.environ.get('EXTRA','{}')))
for m in (pr.s2ag_generator,pr.s2ag_discriminator,pr.trimodal_generator): set_dropout(m,0,0,0)
noise.manual_seed(STEP_SEED)
# oracle alongside
oc=O.ModelCfg(hidden_size=hidden,hidden_size_s2eg=hidden,dropout_prob=0.0)
G=O.recipe_state_dict(O.generator_shapes(oc,n_words,n_spk),s0+1); D=O.recipe_state_dict(O.aff_discriminator_shapes(),s0+2); T3=O.recipe_state_dict(O.trimodal_shapes(oc,n_words,n_spk),s0+4)
gopt,dopt=O.AdamState(),O.AdamState(); scfg=O.StepCfg(**overrides)
class TN(O.Noise):
    def __init__(s,e): super().__init__({'eps':e})
    def dropout(s,name,x,p): return x
names={'speaker':('g_dis','pgt','g_main','g_rand'),'znone':('g_dis','pgt','g_main'),'noreg':('g_dis','pgt','g_main'),'warmup':('pgt','g_main','g_rand')}[variant]
key='aff_encoder.conv4.weight'
for s in range(3):
    perm=torch.from_numpy(g[f's{s}.perm'])
    P.torch.randperm=lambda n,*a,**k: perm
    inp=O.recipe_inputs(B,34,s0+100+s,n_words,n_spk)
    import copy
    Gp={k:v.detach().cpu().clone() for k,v in pr.s2ag_generator.state_dict().items()}
    Dp={k:v.detach().cpu().clone() for k,v in pr.s2ag_discriminator.state_dict().items()}
    Tp={k:v.detach().cpu().clone() for k,v in pr.trimodal_generator.state_dict().items()}
    gopt_p,dopt_p=copy.deepcopy(gopt),copy.deepcopy(dopt)
    # D's Adam state in the product differs slightly; use the oracle's copy (only affects D weights inside the step)
    pr.forward_pass_s2ag(*[inp[k].clone() for k in ('in_text','in_audio','in_mfcc','target','vid')],True)
    eps=torch.from_numpy(g[f's{s}.eps']); bp={n:TN(eps[i]) for i,n in enumerate(names)}
    nz=O.StepNoise(g_dis=bp.get('g_dis'),d_real=O.Noise('off'),d_fake=O.Noise('off'),pgt=bp.get('pgt'),g_main=bp.get('g_main'),d_gen=O.Noise('off'),g_rand=bp.get('g_rand'),perm=perm)
    m,l,gr=O.gan_step(G,D,T3,gopt,dopt,oc,scfg,inp['in_text'],inp['in_audio'],inp['in_mfcc'],inp['target'],inp['vid'],epoch=1,noise=nz)
    named=dict(pr.s2ag_generator.named_parameters())
    worst=sorted(((float((named[k].grad.detach().cpu().double()-gr['G'][k].double()).abs().max()/max(1e-12,float(gr['G'][k].abs().max()))),k) for k in gr['G'] if gr['G'][k] is not None and named[k].grad is not None and not k.endswith('bias')),reverse=True)[:8]
    print('   losses product', {k:round(v,5) for k,v in pr.last_losses.items()}, 'oracle', {k:round(v,5) for k,v in l.items()})
    print('   worst grads', [(round(a,5),k) for a,k in worst])
    _,_,gr2=O.gan_step(Gp,Dp,Tp,gopt_p,dopt_p,oc,scfg,inp['in_text'],inp['in_audio'],inp['in_mfcc'],inp['target'],inp['vid'],epoch=1,noise=nz)
    worst2=sorted(((float((named[k].grad.detach().cpu().double()-gr2['G'][k].double()).abs().max()/max(1e-12,float(gr2['G'][k].abs().max()))),k) for k in gr2['G'] if gr2['G'][k] is not None and named[k].grad is not None and not k.endswith('bias')),reverse=True)[:4]
    print('   worst grads vs oracle FROM THE PRODUCT\'S OWN WEIGHTS', [(round(a,6),k) for a,k in worst2])
    pg=dict(pr.s2ag_generator.named_parameters())[key].grad.detach().cpu().double().reshape(-1)
    og=gr['G'][key].double().reshape(-1)
    d=(pg-og).abs()
    i=int(d.argmax())
    print(f'step {s}: grad max {float(og.abs().max()):.3e} worst abs diff {float(d.max()):.3e} at {i}: product {float(pg[i]):.4e} oracle {float(og[i]):.4e}; smallest |g| {float(og.abs().min()):.3e}')
    w=dict(pr.s2ag_generator.named_parameters())[key].detach().cpu().double().reshape(-1); ow=G[key].double().reshape(-1)
    dw=(w-ow).abs(); j=int(dw.argmax())
    print(f'        weight worst diff {float(dw.max()):.3e} at {j} (lr 5e-4); oracle grad there {float(og[j]):.3e} product grad {float(pg[j]):.3e}')
    sd=pr.s2ag_generator.state_dict()
    w=sorted(((float((sd[k].detach().cpu().double()-G[k].double()).abs().max()/max(1e-12,float(G[k].double().abs().max()))),k) for k in G if '.net.' not in k and not k.endswith('num_batches_tracked') and not k.endswith('.bias')),reverse=True)[:6]
    print('   worst state G', [(round(a,6),k) for a,k in w])
    sd=pr.s2ag_discriminator.state_dict()
    w=sorted(((float((sd[k].detach().cpu().double()-D[k].double()).abs().max()/max(1e-12,float(D[k].double().abs().max()))),k) for k in D if not k.endswith('num_batches_tracked')),reverse=True)[:5]
    print('   worst state D', [(round(a,6),k) for a,k in w])
    for key2 in ('text_encoder.tcn.network.2.conv2.weight_v','aff_encoder.st_gcn2.gcn.conv.weight'):
        w=dict(pr.s2ag_generator.named_parameters())[key2].detach().cpu().double().reshape(-1); ow=G[key2].double().reshape(-1)
        pg2=dict(pr.s2ag_generator.named_parameters())[key2].grad.detach().cpu().double().reshape(-1); og2=gr['G'][key2].double().reshape(-1)
        dw=(w-ow).abs(); j=int(dw.argmax())
        print(f'   {key2}: worst weight diff {float(dw.max()):.3e} at {j}: w {float(w[j]):.5f} vs {float(ow[j]):.5f}; grad product {float(pg2[j]):.3e} oracle {float(og2[j]):.3e}; n(|dw|>1e-5)={int((dw>1e-5).sum())} of {dw.numel()}; exact-zero grads product {int((pg2==0).sum())} oracle {int((og2==0).sum())}')
