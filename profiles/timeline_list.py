import csv, sys, re, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id','?'))) for r in rows]
# last group
groups=[]; cur=[iv[0]]; end=iv[0][1]
for x in iv[1:]:
    if x[0]-end > 20e6: groups.append(cur); cur=[]
    cur.append(x); end=max(end,x[1])
groups.append(cur)
g=groups[-1]; t0=g[0][0]
def short(n):
    m=re.search(r'(ltr_\w+)(<[^>]*>)?', n); return (m.group(1)+(m.group(2) or '')) if m else n[:30]
for s,e,n,q in g: print('%8.3f %8.3f  q%s %s'%((s-t0)/1e6,(e-s)/1e6,q,short(n)))
print('span', (max(e for _,e,_,_ in g)-t0)/1e6)
