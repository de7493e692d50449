import torch, sys
sys.path.insert(0, "/root/repo")
from centerpoly_amd import _C
L=_C.lib(); P=_C.ptr
for (B,ci,co,H,W) in [(4,32,64,512,1024),(4,64,128,256,512),(4,128,256,128,256),(4,256,512,64,128),(4,16,32,1024,2048)]:
    Ho,Wo=(H-1)//2+1,(W-1)//2+1
    w=torch.randn(co,ci,3,3,device="cuda")*0.05; go=torch.randn(B,co,Ho,Wo,device="cuda")
    gx=torch.empty(B,ci,H,W,device="cuda")
    wp=torch.empty(L.cp_conv_mfma_weight_bytes(co,ci,9),dtype=torch.uint8,device="cuda")
    L.cp_conv_mfma_prepare(P(w),co,ci,9,6,P(wp),_C.stream())
    def one(): L.cp_conv3x3_s2_input_grad(P(go),P(wp),None,P(gx),B,ci,H,W,co,_C.stream())
    def lib(): return torch.nn.grad.conv2d_input((B,ci,H,W),w,go,stride=2,padding=1)
    for name,f in (("one",one),("lib",lib)):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print((B,ci,co,H,W), name, "%.1f us" % (e0.elapsed_time(e1)*50), flush=True)
    # weight gradient of the same layer: own kernel vs the library
    x = torch.randn(B, ci, H, W, device="cuda"); gw = torch.zeros_like(w)
    def own_w(): L.cp_conv3x3_s2_wgrad(P(x), P(go), P(gw), B, ci, H, W, co, _C.stream())
    def lib_w(): return torch.nn.grad.conv2d_weight(x, w.shape, go, stride=2, padding=1)
    if L.cp_conv3x3_s2_wgrad_supported(ci, co, H, W) and ci >= 24:
        for name, f in (("wgrad own", own_w), ("wgrad lib", lib_w)):
            for _ in range(3): f()
            torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print((B,ci,co,H,W), name, "%.1f us" % (e0.elapsed_time(e1)*50), flush=True)
