import time, os, sys, torch
sys.path.insert(0, os.getcwd())
nt = int(sys.argv[1]); res = int(sys.argv[2])
torch.set_num_threads(nt)
print("threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), torch.__config__.parallel_info().split("\n")[0:6], flush=True)
t0=time.time()
from oracle.pipeline import install_ip_processors
from oracle.sdxl_unet import UNet2DConditionModel, sdxl_config
with torch.device("meta"):
    u = UNet2DConditionModel(sdxl_config()); install_ip_processors(u, num_tokens=4)
print("meta build", time.time()-t0, flush=True); t0=time.time()
u = u.to_empty(device="cpu").eval()
print("to_empty", time.time()-t0, flush=True); t0=time.time()
with torch.no_grad():
    block = torch.randn(1 << 22)
    for p in u.parameters():
        if p.ndim >= 2:
            flat, sc = p.view(-1), p[0].numel() ** -0.5
            for i in range(0, flat.numel(), block.numel()):
                n = min(block.numel(), flat.numel() - i)
                torch.mul(block[:n], sc, out=flat[i:i + n])
        else:
            p.fill_(1.0)
print("fill", time.time()-t0, flush=True); t0=time.time()
lat=res//8
x = torch.randn(2, 4, lat, lat); ehs = torch.randn(2, 81, 2048)
kw = {"text_embeds": torch.randn(2, 1280), "time_ids": torch.tensor([[res, res, 0, 0, res, res]] * 2, dtype=torch.float32)}
with torch.no_grad():
    u(x, torch.tensor(500.0), ehs, added_cond_kwargs=kw)
print("forward", res, time.time()-t0, flush=True)
