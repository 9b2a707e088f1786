cd $GRAFT_REPO_ROOT
python tools/_diag_fp16.py fp16 stepmode 2>&1 | grep "params equal\|grad equal" | head -24
ACEZ_LIB=$PWD/tools/libacez_r02.so python tools/_diag_fp16.py bf16 /tmp/p_r02.npy 2>&1 | tail -1
python tools/_diag_fp16.py bf16 /tmp/p_new.npy 2>&1 | tail -1
python -c "
import numpy as np
a=np.load('/tmp/p_r02.npy'); b=np.load('/tmp/p_new.npy')
n=2103300
ga,gb=a[:n+4],b[:n+4]; pa,pb=a[n+4:2*n+4],b[n+4:2*n+4]; xa,xb=a[2*n+4:],b[2*n+4:]
print('X equal', np.array_equal(xa,xb), 'stats', ga[n:n+4], gb[n:n+4])
L=8
for l in range(L):
    w=slice(l*262656, l*262656+262144); bb=slice(l*262656+262144,(l+1)*262656)
    print(l, 'gradW diff', int((ga[w]!=gb[w]).sum()), 'gradb diff', int((ga[bb]!=gb[bb]).sum()), 'W diff', int((pa[w]!=pb[w]).sum()), 'b diff', int((pa[bb]!=pb[bb]).sum()))
print('fc3 grad diff', int((ga[L*262656:n]!=gb[L*262656:n]).sum()), 'fc3 diff', int((pa[L*262656:]!=pb[L*262656:]).sum()))
"
