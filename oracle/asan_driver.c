/* ORACLE — TEST INFRASTRUCTURE ONLY. AddressSanitizer/UBSan driver: runs the four oracle query
 * pipelines on generated SF0.01 data in a sanitizer build (CPU only; GPU sanitizers are not
 * available on the pool). Built and run by tests/test_oracle_asan.py via `make -C oracle asan_check`. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "tpchgen.h"
int main(void){
  int64_t num=1,den=100; int64_t no=tpchgen_orders_count(num,den); int64_t n=tpchgen_lineitem_count(num,den,0,no);
  int64_t *ok=malloc(n*8),*ext=malloc(n*8),*disc=malloc(n*8),*tax=malloc(n*8); int32_t *qty=malloc(n*4),*ship=malloc(n*4),*pk=malloc(n*4),*sk=malloc(n*4); uint8_t *rf=malloc(n),*ls=malloc(n);
  tpchgen_lineitem_cols c; memset(&c,0,sizeof c); c.l_orderkey=ok;c.l_extendedprice=ext;c.l_discount=disc;c.l_tax=tax;c.l_quantity=qty;c.l_shipdate=ship;c.l_returnflag=rf;c.l_linestatus=ls;c.l_partkey=pk;c.l_suppkey=sk;
  tpchgen_lineitem(num,den,0,no,&c);
  oracle_lineitem L; memset(&L,0,sizeof L); L.l_quantity=qty;L.l_extendedprice=ext;L.l_discount=disc;L.l_tax=tax;L.l_returnflag=rf;L.l_linestatus=ls;L.l_shipdate=ship;L.l_orderkey=ok;L.l_partkey=pk;L.l_suppkey=sk;L.returnflag_dict=TPCHGEN_RETURNFLAG_DICT;L.linestatus_dict=TPCHGEN_LINESTATUS_DICT;L.n=n;
  oracle_q1_row rows[16]; int g=oracle_q1(&L,tpchgen_days_from_civil(1998,12,1)-112,rows,16);
  char buf[8192]; oracle_q1_text(rows,g,TPCHGEN_RETURNFLAG_DICT,TPCHGEN_LINESTATUS_DICT,buf,sizeof buf); printf("%s",buf);
  odec rev; int rc=oracle_q6(&L,tpchgen_days_from_civil(1994,1,1),tpchgen_days_from_civil(1995,1,1),0.02f,0.04f,24,&rev); oracle_q6_text(&rev,rc,buf,sizeof buf); printf("%s",buf);
  int64_t nc=tpchgen_customer_count(num,den); int32_t *ck=malloc(nc*4); uint8_t *seg=malloc(nc); tpchgen_customer_cols cc; memset(&cc,0,sizeof cc); cc.c_custkey=ck; cc.c_mktsegment=seg; tpchgen_customer(num,den,0,nc,&cc);
  int64_t *ook=malloc(no*8); int32_t *oc=malloc(no*4),*od=malloc(no*4),*op=malloc(no*4); tpchgen_orders_cols o; memset(&o,0,sizeof o); o.o_orderkey=ook;o.o_custkey=oc;o.o_orderdate=od;o.o_shippriority=op; tpchgen_orders(num,den,0,no,&o);
  oracle_orders O={ook,oc,od,op,no}; oracle_customer C={ck,seg,TPCHGEN_MKTSEGMENT_DICT,5,nc};
  oracle_q3_row *q3=malloc(sizeof(oracle_q3_row)*100000); int64_t n3=oracle_q3(&L,&O,&C,"HOUSEHOLD",tpchgen_days_from_civil(1995,3,29),q3,100000); oracle_q3_text(q3,n3,10,buf,sizeof buf); printf("%lld groups\n%s",(long long)n3,buf);
  /* q9 */
  int64_t np=tpchgen_part_count(num,den), ns=tpchgen_supplier_count(num,den);
  int32_t *ppk=malloc(np*4); uint8_t *pc=malloc(np*5); tpchgen_part_cols pcs={ppk,pc}; tpchgen_part(num,den,0,np,&pcs);
  int32_t *off=malloc((np+1)*4); char *bytes=malloc(np*64); int64_t pos=0; for(int64_t i=0;i<np;i++){off[i]=pos; for(int k=0;k<5;k++){const char*w=TPCHGEN_COLORS[pc[5*i+k]]; size_t l=strlen(w); memcpy(bytes+pos,w,l); pos+=l; if(k<4) bytes[pos++]=' ';}} off[np]=pos;
  int32_t *psp=malloc(np*16),*pss=malloc(np*16); int64_t *psc=malloc(np*32); tpchgen_partsupp_cols psx={psp,pss,psc}; tpchgen_partsupp(num,den,0,np,&psx);
  int32_t *ssk=malloc(ns*4),*sn=malloc(ns*4); tpchgen_supplier_cols sx={ssk,sn}; tpchgen_supplier(num,den,0,ns,&sx);
  oracle_part P={ppk,off,bytes,np}; oracle_partsupp PS={psp,pss,psc,np*4}; oracle_supplier S={ssk,sn,ns};
  oracle_q9_row q9[512]; int64_t n9=oracle_q9(&L,&O,&P,&PS,&S,"%pink%",q9,512); char big[65536]; oracle_q9_text(q9,n9,TPCHGEN_NATION_NAMES,big,sizeof big); printf("%lld q9 groups\n",(long long)n9);
  /* the shapes added after the four queries: OR / IN lists, CASE, filtered (DISTINCT) sinks */
  { ocol sc; memset(&sc,0,sizeof sc); sc.type=OT_INT32; sc.data=qty;
    oconst k1; memset(&k1,0,sizeof k1); k1.type=OT_INT32; k1.i=7; oconst k2=k1; k2.i=49; oconst ks[2]={k1,k2};
    ocol cols2[2]={sc,sc}; int32_t ops[2]={OP_EQ,OP_EQ}; int64_t *selo=malloc(n*8);
    int64_t m=oracle_select_or(cols2,ops,ks,2,NULL,n,selo); printf("%lld rows IN (7,49)\n",(long long)m); free(selo);
    ocol dc[2]; memset(dc,0,sizeof dc); dc[0].type=OT_DECIMAL; dc[0].scale=2; dc[0].data=ext; dc[1].type=OT_DECIMAL; dc[1].scale=2; dc[1].data=disc;
    ocol wc; memset(&wc,0,sizeof wc); wc.type=OT_DATE; wc.data=ship; oconst wk; memset(&wk,0,sizeof wk); wk.type=OT_DATE; wk.i=tpchgen_days_from_civil(1995,6,17);
    orpn tp[5]; memset(tp,0,sizeof tp); tp[0].op=OX_COL; tp[0].col=0; tp[1].op=OX_CONST_INT; tp[1].ival=1; tp[2].op=OX_COL; tp[2].col=1; tp[3].op=OX_SUB; tp[4].op=OX_MUL;
    orpn ep[1]; memset(ep,0,sizeof ep); ep[0].op=OX_CONST_DEC; ep[0].ival=0; ep[0].scale=4;
    odec *co=malloc(sizeof(odec)*n); uint8_t *cn=malloc(n);
    int rcx=oracle_case_decimal(dc,&wc,OP_LT,&wk,tp,5,ep,1,n,co,cn); printf("case rc %d\n",rcx); free(co); free(cn);
    ocol kp; memset(&kp,0,sizeof kp); kp.type=OT_INT32; ocol ap; memset(&ap,0,sizeof ap); ap.type=OT_INT32;
    oaggspec sp[2]={{OA_COUNT,0},{OA_SUM,0}}; oagg *t=oracle_agg_create(&kp,1,&ap,sp,2);
    for(int64_t b=0;b<n && b<20480;b+=2048){ int64_t cnt=n-b<2048?n-b:2048; ocol kc=kp; kc.data=pk+b; ocol ac=ap; ac.data=qty+b; oracle_agg_sink_filtered(t,&kc,&ac,NULL,cnt,b?1u:2u); }
    printf("%lld filtered groups\n",(long long)oracle_agg_count(t)); oracle_agg_free(t);
    ocol sk[3]; memset(sk,0,sizeof sk); sk[0].type=OT_DECIMAL; sk[0].scale=2; sk[0].data=ext; sk[1].type=OT_DATE; sk[1].data=ship; sk[2].type=OT_INT32; sk[2].data=qty;
    int32_t dsc[3]={1,0,1}; int64_t ns=n<20000?n:20000; int64_t *so=malloc(ns*8); int32_t kl=0; uint8_t *kb=malloc((size_t)ns*40);
    int rs=oracle_sort_rows(sk,dsc,3,NULL,ns,so,&kl,kb); printf("sort rc %d key bytes %d first row %lld\n",rs,kl,(long long)so[0]); free(so); free(kb); }
  return 0; }
