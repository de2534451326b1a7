import os, sys
sys.path.insert(0, "/root/repo/faster-orefsdet_amd")
import torch, orehip
import torch.nn.functional as F
L = orehip.lib(); dev = torch.device("cuda")
for name, B, H, W, Cin, Cout in [("stem2",1,320,320,64,64),("s2l0",1,160,160,128,64),("s2l1",1,160,160,64,64),("out3",1,80,80,128,128),("stem2x16",16,320,320,64,64)]:
    x = torch.zeros(B*H*W+2, Cin, dtype=torch.bfloat16, device=dev); x[:-2] = torch.randn(B*H*W, Cin, device=dev).to(torch.bfloat16)
    xd = x[:-2].view(B,H,W,Cin)
    wt = (torch.randn(Cout,Cin,3,3)/(Cin*9)**0.5).to(torch.bfloat16)
    w = orehip.pack_conv_weight_bf16(wt.float()).to(dev)
    sh = torch.randn(Cout, device=dev)
    out = torch.empty(B,H,W,Cout,dtype=torch.bfloat16,device=dev)
    row=[]
    for mode in (0,1):
        L.ore_conv_set_plan_override(-8, mode, 0,0,0)
        f=lambda: orehip.conv2d(xd,w,Cout,3,1,shift=sh,relu_cout=Cout,out=out)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        n=30 if B==1 else 5
        a.record()
        for _ in range(n): f()
        b.record(); torch.cuda.synchronize()
        row.append(a.elapsed_time(b)*1e3/n)
        if mode==1 and B==1:
            ref=F.relu(F.conv2d(xd.float().permute(0,3,1,2).cpu(), wt.float(), sh.cpu(),1,1))
            got=out.float().cpu().permute(0,3,1,2)
            err=float((got-ref).abs().max()/ref.abs().max())
    print(name, "gs/kw %.1f us   ws %.1f us   err %.2e" % (row[0], row[1], err), flush=True)
