// tools/micro/need_model.c — CPU model behind DESIGN.md section 4 "need-gated tables" (TEST / ANALYSIS INFRASTRUCTURE, not part of the library).
// Runs the reference's double-fast parse (hashLog 14 / chainLog 13 / minMatch 5, one 64 KiB block per frame) over a file of concatenated frames with
// two sets of tables side by side: the full ones, and "gated" ones that are read only at positions whose key some other position of the frame
// shares (long: the 8 bytes; short: bucket + 4 bytes — exact keys here, Bloom filters in zj_need.h) and written only in buckets such a position
// reads.  Prints the request counts and checks at every probe that the gated answer cannot change a decision.
//   gcc -O2 -o need_model tools/micro/need_model.c && python -c "import __graft_entry__ as e; open('/tmp/corpus.bin','wb').write(e.load_package().synth_host(65536,0,256))" && ./need_model /tmp/corpus.bin 65536
//   bench set (256 frames): 4.83 -> 1.83 random table requests per searched position, 0 decision differences; per class: searched positions 0.09 / 0.15 / 0.88 / 0.02 per byte.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
static u32 rd32(const u8* p){u32 v;memcpy(&v,p,4);return v;} static u64 rd64(const u8* p){u64 v;memcpy(&v,p,8);return v;}
static u32 hash8(const u8* p,u32 h){return (u32)((rd64(p)*0xCF1BBCDCB7A56463ULL)>>(64-h));}
static u32 hash5(const u8* p,u32 h){return (u32)(((rd64(p)<<24)*889523592379ULL)>>(64-h));}
static u32 count_match(const u8* a,const u8* b,const u8* end){u32 n=0;while(a+n<end&&a[n]==b[n])n++;return n;}
#define HL 14
#define HS 13
static u32 hashLong[1<<HL], hashSmall[1<<HS], gLong[1<<HL], gSmall[1<<HS];
static u8 needL[1<<17], needS[1<<17], bneedL[1<<HL], bneedS[1<<HS];
static u64 nSearch, rdL, rdS, grdL, grdS, wrL, wrS, gwrL, gwrS, bad, nSeq, bytesTotal;
static const u8* base; static u32 N;
typedef struct { u64 k; u32 p; } KP;
static int cmp(const void*a,const void*b){ const KP*x=a,*y=b; return x->k<y->k?-1:(x->k>y->k?1:0); }
static KP kp[1<<17];
static void prep(void){
    u32 n = N>=8? N-7:0;   // positions with 8 readable bytes (the parser never probes beyond ilimit = N-8)
    for(u32 p=0;p<n;p++){kp[p].k=rd64(base+p);kp[p].p=p;} qsort(kp,n,sizeof(KP),cmp);
    memset(needL,0,sizeof needL); memset(needS,0,sizeof needS); memset(bneedL,0,sizeof bneedL); memset(bneedS,0,sizeof bneedS);
    for(u32 i=0;i<n;i++){ int dup=(i>0&&kp[i-1].k==kp[i].k)||(i+1<n&&kp[i+1].k==kp[i].k); if(dup){needL[kp[i].p]=1; bneedL[hash8(base+kp[i].p,HL)]=1;} }
    for(u32 p=0;p<n;p++){kp[p].k=((u64)hash5(base+p,HS)<<32)|rd32(base+p);kp[p].p=p;} qsort(kp,n,sizeof(KP),cmp);
    for(u32 i=0;i<n;i++){ int dup=(i>0&&kp[i-1].k==kp[i].k)||(i+1<n&&kp[i+1].k==kp[i].k); if(dup){needS[kp[i].p]=1; bneedS[hash5(base+kp[i].p,HS)]=1;} }
}
static void insL(u32 pos){ u32 h=hash8(base+pos,HL); hashLong[h]=pos+1; wrL++; if(bneedL[h]){gLong[h]=pos+1;gwrL++;} }
static void insS(u32 pos){ u32 h=hash5(base+pos,HS); hashSmall[h]=pos+1; wrS++; if(bneedS[h]){gSmall[h]=pos+1;gwrS++;} }
static u32 readL(u32 pos){ u32 h=hash8(base+pos,HL); u32 v=hashLong[h]; rdL++; if(needL[pos]){grdL++; if(gLong[h]!=v) bad++; return v;} if (v && rd64(base+v-1)==rd64(base+pos)) bad++; return 0; }
static u32 readS(u32 pos){ u32 h=hash5(base+pos,HS); u32 v=hashSmall[h]; rdS++; if(needS[pos]){grdS++; if(gSmall[h]!=v) bad++; return v;} if (v && rd32(base+v-1)==rd32(base+pos)) bad++; return 0; }
static void frame(const u8* src, u32 srcSize){
    base=src; N=srcSize; prep(); memset(hashLong,0,sizeof hashLong); memset(hashSmall,0,sizeof hashSmall); memset(gLong,0,sizeof gLong); memset(gSmall,0,sizeof gSmall);
    const u8* const istart=src; const u8* const iend=src+srcSize; const u8* const ilimit=iend-8; const u8* anchor=istart; const u8* ip=istart; const u8* ip1;
    u32 off1=1,off2=4,mLength,offset,curr=0,step,el0,el1; const u8* nextStep; const u8* matchs0; const u8* matchl0;
    ip+=1; {u32 maxRep=(u32)(ip-istart); if(off2>maxRep)off2=0; if(off1>maxRep)off1=0;}
    for(;;){
        step=1; nextStep=ip+256; ip1=ip+step; if(ip1>ilimit) return;
        el0=readL((u32)(ip-istart));
        do { nSearch++;
            u32 const es0=readS((u32)(ip-istart));
            curr=(u32)(ip-istart); insL(curr); insS(curr);
            if((off1>0)&(rd32(ip+1-off1)==rd32(ip+1))){mLength=count_match(ip+1+4,ip+1+4-off1,iend)+4;ip++;nSeq++;goto _stored;}
            if(el0&&rd64(istart+el0-1)==rd64(ip)){matchl0=istart+el0-1;mLength=count_match(ip+8,matchl0+8,iend)+8;offset=(u32)(ip-matchl0);
                while(((ip>anchor)&(matchl0>istart))&&(ip[-1]==matchl0[-1])){ip--;matchl0--;mLength++;} goto _found;}
            el1=readL((u32)(ip1-istart));
            if(es0&&rd32(istart+es0-1)==rd32(ip)){matchs0=istart+es0-1;goto _next_long;}
            if(ip1>=nextStep){step++;nextStep+=256;}
            ip=ip1;ip1+=step;el0=el1;
        } while(ip1<=ilimit);
        return;
_next_long:
        mLength=count_match(ip+4,matchs0+4,iend)+4;offset=(u32)(ip-matchs0);
        if((el1>1)&&(rd64(istart+el1-1)==rd64(ip1))){const u8* m1=istart+el1-1;u32 l1=count_match(ip1+8,m1+8,iend)+8;if(l1>mLength){ip=ip1;mLength=l1;offset=(u32)(ip-m1);matchs0=m1;}}
        while(((ip>anchor)&(matchs0>istart))&&(ip[-1]==matchs0[-1])){ip--;matchs0--;mLength++;}
_found:
        off2=off1;off1=offset;nSeq++;
        if(step<4) insL((u32)(ip1-istart));
_stored:
        ip+=mLength;anchor=ip;
        if(ip<=ilimit){ u32 const ins=curr+2; insL(ins); insL((u32)(ip-2-istart)); insS(ins); insS((u32)(ip-1-istart));
            while((ip<=ilimit)&&((off2>0)&(rd32(ip)==rd32(ip-off2)))){u32 r=count_match(ip+4,ip+4-off2,iend)+4;u32 t=off2;off2=off1;off1=t;insS((u32)(ip-istart));insL((u32)(ip-istart));ip+=r;anchor=ip;nSeq++;} }
    }
}
int main(int argc,char**argv){ FILE*f=fopen(argv[1],"rb"); u32 fs=atoi(argv[2]); u8* buf=malloc(fs+64); u32 n=0; memset(buf,0,fs+64);
    while(fread(buf,1,fs,f)==fs){ frame(buf,fs); n++; bytesTotal+=fs; }
    printf("frames %u, searched positions %llu (%.3f per byte), sequences %llu\n",n,(unsigned long long)nSearch,(double)nSearch/bytesTotal,(unsigned long long)nSeq);
    printf("table reads  now: long %llu short %llu | gated: long %llu (%.1f%%) short %llu (%.1f%%)\n",(unsigned long long)rdL,(unsigned long long)rdS,(unsigned long long)grdL,100.0*grdL/rdL,(unsigned long long)grdS,100.0*grdS/rdS);
    printf("table writes now: long %llu short %llu | gated: long %llu (%.1f%%) short %llu (%.1f%%)\n",(unsigned long long)wrL,(unsigned long long)wrS,(unsigned long long)gwrL,100.0*gwrL/wrL,(unsigned long long)gwrS,100.0*gwrS/wrS);
    printf("random table requests per searched position: now %.2f, gated %.2f (x%.1f fewer)   decision differences: %llu\n",(double)(rdL+rdS+wrL+wrS)/nSearch,(double)(grdL+grdS+gwrL+gwrS)/nSearch,(double)(rdL+rdS+wrL+wrS)/(double)(grdL+grdS+gwrL+gwrS),(unsigned long long)bad);
    return 0; }
