import torch, time
dev = torch.device("cuda:0")
x = torch.randn(1, 2304, 180, 180, device=dev, requires_grad=True)
w = torch.randn(108, 64, 3, 3, device=dev, requires_grad=True)
for cl in (False, True):
    xi = x.detach().clone()
    if cl:
        xi = xi.contiguous(memory_format=torch.channels_last)
    xi.requires_grad_(True)
    for it in range(3):
        y = torch.nn.functional.conv2d(xi, w, None, 1, 1, 1, 36)
        y.sum().backward()
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(5):
        y = torch.nn.functional.conv2d(xi, w, None, 1, 1, 1, 36)
        y.sum().backward()
    torch.cuda.synchronize()
    print("channels_last", cl, "grouped conv fwd+bwd ms", (time.time() - t0) / 5 * 1e3)
