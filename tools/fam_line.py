import sys,json
d=json.loads(sys.stdin.read()); f=d["roofline"]["families"]
print(d["ms_per_step"], ' '.join('%s=%.0f/%d'%(k.replace('conv3x3_','').replace('_kernel',''),v['us_per_step'],v.get('launches',0)) for k,v in sorted(f.items(), key=lambda kv:-kv[1]['us_per_step'])[:12]))
