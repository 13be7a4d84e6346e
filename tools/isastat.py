import re,sys
s=open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m=re.search(r'^(%s): .*?\n(.*?)\.end_amdhsa_kernel'%re.escape(name), s, re.S|re.M)
    if not m: print('no',name); continue
    body=m.group(2)
    print(name[:80])
    out=[]
    for k in ['v_mfma','ds_read_b128','ds_write','global_load_dwordx4','global_load_lds','buffer_store','global_store','v_accvgpr_read','v_accvgpr_write','scratch_','s_waitcnt','s_cbranch','s_nop','s_barrier']:
        out.append('%s %d'%(k, len(re.findall(k, body))))
    print('   ', ', '.join(out))
    out=[]
    for k in ['next_free_vgpr','accum_offset','next_free_sgpr','private_segment_fixed_size']:
        mm=re.search(r'\.amdhsa_%s (\d+)'%k, body); out.append('%s %s'%(k, mm.group(1) if mm else None))
    print('   ', ', '.join(out), 'lines', body.count('\n'))
