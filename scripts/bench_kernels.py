import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(d['ms_per_step'], d['value'])
for k in sorted(d['kernels'], key=lambda r:-r['us']*r['launches'])[:26]:
    print('%-42s launches %5.1f  us %8.2f  frac %.3f' % (k['kernel'], k['launches']/d['kernel_timing']['steps'], k['us'], k['frac']))
