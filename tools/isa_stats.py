import sys,re,collections
fn, pat = sys.argv[1], sys.argv[2]
lines=open(fn).read().splitlines()
# find function by mangled-name substring
start=None
for i,l in enumerate(lines):
    if l.startswith(pat) and ': ;' in l:
        start=i;break
end=next(i for i in range(start,len(lines)) if lines[i].strip().startswith('s_endpgm'))
body=lines[start:end]
# find the main loop: largest span between a label and a backward branch to it
labels={re.match(r'^(\.LBB\d+_\d+):',l).group(1):i for i,l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:',l)}
best=(0,0,0)
for i,l in enumerate(body):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l) or re.search(r's_branch (\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        if i-labels[m.group(1)]>best[0]: best=(i-labels[m.group(1)],labels[m.group(1)],i)
_,a,b=best
c=collections.Counter()
for l in body[a:b]:
    l=l.strip()
    if not l or l.startswith(';') or l.startswith('.'): continue
    op=l.split()[0]
    c[op]+=1
tot=sum(c.values())
print(fn,"loop lines",b-a,"instrs",tot)
grp=collections.Counter()
for op,n in c.items():
    if op.startswith('v_pk'): grp['v_pk']+=n
    elif op.startswith('v_'): grp['v_other']+=n
    elif op.startswith('ds_'): grp[op]+=n
    elif op.startswith('s_waitcnt'): grp['s_waitcnt']+=n
    elif op.startswith('s_barrier'): grp['s_barrier']+=n
    elif op.startswith('global_') or op.startswith('buffer_'): grp[op]+=n
    elif op.startswith('s_'): grp['s_other']+=n
    else: grp[op]+=n
for k,v in sorted(grp.items(), key=lambda x:-x[1]): print("  %-24s %d"%(k,v))
print("  top v_other:", [ (k,v) for k,v in c.most_common(40) if k.startswith('v_') and not k.startswith('v_pk')][:8])
