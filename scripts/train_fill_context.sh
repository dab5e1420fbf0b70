#!/bin/bash
# which kernels surround the small torch fill launches of the training step (kernel trace, last step only)
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_fill -- python scripts/train_step.py --steps 2 --warmup 2 > /tmp/kt_fill.log 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/kt_fill/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    if 'FillFunctor<float>' in n: return 'FillFunctor<float>'
    m = re.search(r'(k_\w+|FillFunctor<\w+>|multi_tensor_apply_kernel|copyBuffer|fillBuffer|\w+_kernel\w*)', n)
    return m.group(1) if m else n[:40]
names = [short(r['Kernel_Name']) for r in rows]
cs = [i for i, n in enumerate(names) if n == 'k_centers']
seq = names[cs[-2]:cs[-1]]                                  # one whole step, from one grouping stage to the next
ctx = collections.Counter()
for i, n in enumerate(seq):
    if n.startswith('FillFunctor<float>'):
        ctx[(seq[i - 1] if i else '-', seq[i + 1] if i + 1 < len(seq) else '-')] += 1
print("fills in the last step:", sum(ctx.values()))
for k, c in ctx.most_common(12):
    print("%4d   after %-32s before %s" % (c, k[0], k[1]))
PY
