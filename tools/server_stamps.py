"""Per-block phase stamps of one evaluation-server round (development aid): writes gpurun_out/stamps.bin and prints the distribution."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000); src = clouds.source_from_target(tgt, 100000)
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
g.setInputTarget(tgt); g.setInputSource(src)
os.environ["NDT_DIAG_DUMP"] = "gpurun_out/stamps.bin"
print(g.diag_server_roundtrip(np.zeros(6)))
d = np.fromfile("gpurun_out/stamps.bin", dtype=np.uint64).astype(np.int64)
NB = int(os.environ.get("NDT_STAMP_BLOCKS", "256"))
t0 = d[0]; got = (d[8:8+2*NB:2]-t0)*0.01; tk = (d[9:9+2*NB:2]-t0)*0.01
dur = tk-got
print("got  ", np.round(got[:32],2))
print("dur by block (first 64)", np.round(dur[:64],1))
print("dur percentiles", np.percentile(dur,[0,10,50,90,99,100]).round(2))
for x in range(8):
    print("xcd", x, "median dur %.2f max %.2f  ticket max %.2f" % (np.median(dur[x::8]), dur[x::8].max(), tk[x::8].max()))
o = np.argsort(tk)
print("slowest blocks", o[-12:], np.round(tk[o[-12:]],2))
nb = len(dur)
fine = (d[8+2*1024:8+2*1024+8*1024].reshape(1024,8)[:NB,:5]-t0)*0.01
allst = np.column_stack([got[:NB], fine, tk[:NB]])
names = ["got","tables","body","fold","sync","stored","ticket"]
dl = np.diff(allst, axis=1)
for i in range(6):
    print("%-7s->%-7s median %.2f p90 %.2f max %.2f" % (names[i], names[i+1], np.median(dl[:,i]), np.percentile(dl[:,i],90), dl[:,i].max()))
print("summer starts (latest) %.2f  published (latest) %.2f  rows sent: median %.2f max %.2f" % ((d[2]-t0)*0.01, (d[3]-t0)*0.01, np.median(tk), tk.max()))
