"""Instruction census of the inner loops (Depth = 2, >= 64 MFMAs) of one kernel in a hipcc -S listing:  python tools/isa_loop_census.py listing.s [mangled-kernel-label:]"""
import collections,re,sys
fn=sys.argv[1]; key=sys.argv[2] if len(sys.argv)>2 else '_ZN4dsvt16conv_wide_kernelILi8ELi8ELi36ELi4ELi2ELi2ELb0ELb1ELb0EEEvNS_8ConvArgsEPKDF16_S3_iiii:'
txt=open(fn).read()
i=txt.find('\n'+key); j=txt.find('.Lfunc_end',i)
L=txt[i:j].split('\n')
# inner loops: label lines followed (within 3 lines) by 'Inner Loop Header'
res=[]
for k,l in enumerate(L):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m and any('Inner Loop Header' in x for x in L[k:k+4]):
        lab=m.group(1)
        backs=[q for q,x in enumerate(L) if q>k and re.search(r's_(c)?branch\w*\s+'+re.escape(lab)+r'\b',x)]
        if backs: res.append((k,backs[-1],lab))
for a,b,lab in res:
    c=collections.Counter()
    for l in L[a:b+1]:
        l=l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
        c[l.split()[0]]+=1
    nm=sum(v for k,v in c.items() if 'mfma' in k)
    if nm<64: continue
    print(fn.split('/')[-1],lab,'total',sum(c.values()),'mfma',nm,'valu',sum(v for k,v in c.items() if k.startswith('v_') and 'mfma' not in k),'salu',sum(v for k,v in c.items() if k.startswith('s_')),'br',sum(v for k,v in c.items() if 'cbranch' in k),'wait',c['s_waitcnt'],'mov64',c['v_mov_b64_e32'],'rdlane',c['v_readlane_b32']+c['v_writelane_b32'],'scratch',sum(v for k,v in c.items() if 'scratch' in k),'ds_read',c['ds_read_b128'],'vmem',c['global_load_lds_dwordx4']+c['buffer_load_dwordx4'])
