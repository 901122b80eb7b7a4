import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'),'tests'))
from conftest import load_golden
from oracle import ref_cpu as R
from dmvae_amd import ops
from dmvae_amd.models.flux_ae import ResnetBlock
l2 = lambda a,b: ((a.double()-b.double()).norm()/b.double().norm()).item()
g = load_golden("resblock_same"); p = g.sub("p.")
m = ResnetBlock(64,64); m.load_state_dict(p); m = m.cuda()
x = g.t("x").cuda()
y = m(x).float().cpu()
with torch.no_grad():
    yq = R.resnet_block(R.bf16_round(g.t("x")), p, "", R.bf16_round)
    y0 = R.resnet_block(g.t("x"), p, "")
print("block: hip vs oracleQ", l2(y,yq), "hip vs f32", l2(y,y0), "oracleQ vs f32", l2(yq,y0))
# step by step
xb = x.permute(0,2,3,1).contiguous().bfloat16()
st = ops.groupnorm_stats(xb); a1 = ops.groupnorm_apply(xb, st, m.norm1.weight, m.norm1.bias, True)
xq = R.bf16_round(g.t("x"))
a1o = R.bf16_round(R.swish(R.group_norm(xq, p["norm1.weight"], p["norm1.bias"])))
print("a1 vs oracle:", l2(a1.float().cpu().permute(0,3,1,2), a1o), "n mismatching elems", (a1.float().cpu().permute(0,3,1,2)!=a1o).sum().item(), a1o.numel())
from dmvae_amd.functional import packed
h1 = ops.conv2d_nhwc(a1, packed(m.conv1.weight), m.conv1.bias, ks=3)
h1o = R.conv2d(a1o, p, "conv1", R.bf16_round)
d = (h1.float().cpu().permute(0,3,1,2)!=h1o)
print("h1 vs oracle:", l2(h1.float().cpu().permute(0,3,1,2), h1o), "mismatch", d.sum().item(), h1o.numel())
h1f = ops.conv2d_nhwc(a1, packed(m.conv1.weight), m.conv1.bias, ks=3, out_f32=True)
h1of = torch.nn.functional.conv2d(a1.float().cpu().permute(0,3,1,2), p["conv1.weight"].bfloat16().float(), p["conv1.bias"], padding=1)
print("h1 f32 (same a1 input) vs torch:", l2(h1f.cpu().permute(0,3,1,2), h1of))
