import sys, os
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
    x = torch.zeros(1 << 20, device="cuda")
sys.argv = sys.argv[:1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_probe.py")).read())
