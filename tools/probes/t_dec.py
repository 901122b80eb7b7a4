import sys, os, torch
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,'tests'))
from conftest import load_golden
from oracle import ref_cpu as R
from oracle.detweights import det_tensor
from dmvae_amd import functional as Fn
from dmvae_amd.models.flux_ae import Decoder
l2 = lambda a,b: ((a.double()-b.double()).norm()/b.double().norm()).item()
g = load_golden("decoder_small")
dec = Decoder(ch=32, out_ch=3, ch_mult=(1,2,4,4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16); dec.post_init(32)
params = {k: det_tensor(k, v.shape, 12) for k, v in dec.state_dict().items()}
dec.load_state_dict(params); dec = dec.cuda()
Q = R.bf16_round
def cmp(name, h, ho):
    hh = h.float().cpu().permute(0,3,1,2)
    print(f"{name:28s} l2 {l2(hh, ho):.2e} mismatches {(hh!=ho).sum().item()}/{ho.numel()}")
with torch.no_grad():
    z = g.t("z")
    h = Fn.to_nhwc_bf16(z.cuda()); ho = Q(z)
    h = dec.conv_in[0].forward_nhwc(h); ho = R.upsample(ho, params, "conv_in.0.", Q); cmp("conv_in.0", h, ho)
    h = dec.conv_in[1].forward_nhwc(h); ho = R.conv2d(ho, params, "conv_in.1", Q); cmp("conv_in.1", h, ho)
    h = dec.mid.block_1.forward_nhwc(h); ho = R.resnet_block(ho, params, "mid.block_1.", Q); cmp("mid.block_1", h, ho)
    h = dec.mid.attn_1.forward_nhwc(h); ho = R.attn_block(ho, params, "mid.attn_1.", Q); cmp("mid.attn_1", h, ho)
    h = dec.mid.block_2.forward_nhwc(h); ho = R.resnet_block(ho, params, "mid.block_2.", Q); cmp("mid.block_2", h, ho)
    for lvl in reversed(range(4)):
        for b in range(3):
            h = dec.up[lvl].block[b].forward_nhwc(h); ho = R.resnet_block(ho, params, f"up.{lvl}.block.{b}.", Q); cmp(f"up.{lvl}.block.{b}", h, ho)
        if lvl:
            h = dec.up[lvl].upsample.forward_nhwc(h); ho = R.upsample(ho, params, f"up.{lvl}.upsample.", Q); cmp(f"up.{lvl}.upsample", h, ho)
