import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/omnihd-scenes_amd", "/root/repo/tests"]
os.chdir("/root/repo")
import torch
import test_detector_gpu as T
res = {}
for pol in ("miopen", "split"):
    os.environ["OMNIHD_FP32_CONV"] = pol
    res[pol] = T._run("cuda:0", use_oracle=False, variant="rcfusion")
res["cpu"] = T._run("cpu", use_oracle=True, variant="rcfusion")
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
for k in ("depth", "bev", "cls", "reg"):
    print(k, "split-vs-cpu %.2e  miopen-vs-cpu %.2e  split-vs-miopen %.2e" % (rel(res["split"][k], res["cpu"][k]), rel(res["miopen"][k], res["cpu"][k]), rel(res["split"][k], res["miopen"][k])))
worst = []
for n in res["cpu"]["grads"]:
    a, b, c = res["split"]["grads"][n], res["miopen"]["grads"][n], res["cpu"]["grads"][n]
    worst.append((rel(a, c), rel(b, c), n))
worst.sort(reverse=True)
for w in worst[:25]:
    print("%.2e (split)  %.2e (miopen)  %s" % w)
