// tools/micro/prevlink_model.c — CPU model of an alternative that was NOT built (TEST / ANALYSIS INFRASTRUCTURE): double-fast candidates from
// precomputed "previous position in this bucket" links plus inserted-flags instead of hash tables.  It checks, probe by probe, that the first
// *inserted* position on a bucket's chain is exactly what the reference's table holds (including the one out-of-order insert, position ip1 before
// curr + 2), and counts the chain-walk reads that would replace the table traffic: 2.09 random reads per searched position against 4.8 requests,
// 0.59 with a same-hash filter — but every walk step is a dependent round trip on the critical frames' path, which is why zj_need.h gates the
// tables instead of replacing them.  Build / run like need_model.c.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
static u32 rd32(const u8* p){u32 v;memcpy(&v,p,4);return v;} static u64 rd64(const u8* p){u64 v;memcpy(&v,p,8);return v;}
static u32 hash8(const u8* p,u32 h){return (u32)((rd64(p)*0xCF1BBCDCB7A56463ULL)>>(64-h));}
static u32 hash5(const u8* p,u32 h){return (u32)(((rd64(p)<<24)*889523592379ULL)>>(64-h));}
static u32 hashS(const u8* p,u32 h,u32 mls){ if(mls==5) return hash5(p,h); if (mls==4) return (rd32(p)*2654435761U)>>(32-h); abort(); }
static u32 tag8(const u8* p,u32 h){return (u32)(((rd64(p)*0xCF1BBCDCB7A56463ULL)>>(64-h-15))&0x7FFF);}
static u32 count_match(const u8* a,const u8* b,const u8* end){u32 n=0;while(a+n<end&&a[n]==b[n])n++;return n;}
#define HL 14
#define HS 13
#define MLS 5
static u32 hashLong[1<<HL], hashSmall[1<<HS];
// shadow structures over ALL positions
static u32 allL[1<<HL], allS[1<<HS];          // last position (any) per bucket, +1
static u32 prevL[1<<17], prevS[1<<17];         // +1 encoded
static u8 insL[1<<17], insS[1<<17];
static u64 needL, needS, needReadsL, needReadsS; static u64 nSearch, nReadL, nReadS, stepsL, stepsS, walkReadsL, walkReadsS, mismatch, fetchL, fetchS, nSeq, emptyL, emptyS, longWalks;
static const u8* base;
static u32 tag5(const u8* p,u32 h){return (u32)((((rd64(p)<<24)*889523592379ULL)>>(64-h-15))&0x7FFF);}
static void needWalkL(u32 p){ u32 t=tag8(base+p,HL); u32 q=prevL[p]; int any=0; while(q){ if (tag8(base+q-1,HL)==t){any=1;break;} q=prevL[q-1]; }
    if(!any) return; needL++; q=prevL[p]; while(q){ needReadsL++; if (insL[q-1]) break; q=prevL[q-1]; } }
static void needWalkS(u32 p){ u32 t=tag5(base+p,HS); u32 q=prevS[p]; int any=0; while(q){ if (tag5(base+q-1,HS)==t){any=1;break;} q=prevS[q-1]; }
    if(!any) return; needS++; q=prevS[p]; while(q){ needReadsS++; if (insS[q-1]) break; q=prevS[q-1]; } }
static u32 chainL(u32 p, u32* reads){ // candidate for long table at position p: first inserted on bucket chain; returns +1 encoded
    u32 q = prevL[p]; u32 r = 0;
    while (q) { r++; if (insL[q-1]) break; q = prevL[q-1]; }
    *reads = r; return q; }
static u32 chainS(u32 p, u32* reads){ u32 q = prevS[p]; u32 r = 0; while (q) { r++; if (insS[q-1]) break; q = prevS[q-1]; } *reads = r; return q; }
static void insertL(u32 pos){ hashLong[hash8(base+pos,HL)] = pos+1; insL[pos]=1; }
static void insertS(u32 pos){ hashSmall[hashS(base+pos,HS,MLS)] = pos+1; insS[pos]=1; }
static u32 readL(u32 pos){ needWalkL(pos); u32 v = hashLong[hash8(base+pos,HL)]; u32 r; u32 c = chainL(pos,&r); nReadL++; walkReadsL += r; if (r>8) longWalks++; if (!c) emptyL++; if (c!=v) mismatch++; return v; }
static u32 readS(u32 pos){ needWalkS(pos); u32 v = hashSmall[hashS(base+pos,HS,MLS)]; u32 r; u32 c = chainS(pos,&r); nReadS++; walkReadsS += r; if (r>8) longWalks++; if (!c) emptyS++; if (c!=v) mismatch++; return v; }
static void frame(const u8* src, u32 srcSize){
    base = src; memset(hashLong,0,sizeof hashLong); memset(hashSmall,0,sizeof hashSmall); memset(allL,0,sizeof allL); memset(allS,0,sizeof allS);
    memset(insL,0,sizeof insL); memset(insS,0,sizeof insS);
    for (u32 p=0;p+8<=srcSize;p++){ u32 h=hash8(src+p,HL); prevL[p]=allL[h]; allL[h]=p+1; h=hashS(src+p,HS,MLS); prevS[p]=allS[h]; allS[h]=p+1; }
    const u8* const istart = src; const u8* const iend = src + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip = istart; const u8* ip1;
    u32 off1 = 1, off2 = 4, mLength, offset, curr = 0, step, el0, el1; const u8* nextStep; const u8* matchs0; const u8* matchl0;
    ip += 1; { u32 maxRep=(u32)(ip-istart); if (off2>maxRep) off2=0; if (off1>maxRep) off1=0; }
    for(;;){
        step=1; nextStep=ip+256; ip1=ip+step; if (ip1>ilimit) return;
        el0 = readL((u32)(ip-istart));
        do {
            nSearch++;
            u32 const es0 = readS((u32)(ip-istart));
            curr=(u32)(ip-istart); insertL(curr); insertS(curr);
            if ((off1>0)&(rd32(ip+1-off1)==rd32(ip+1))) { mLength=count_match(ip+1+4,ip+1+4-off1,iend)+4; ip++; nSeq++; goto _stored; }
            if (el0) { if (tag8(istart+el0-1,HL)==tag8(ip,HL)) fetchL++; }
            if (el0 && rd64(istart+el0-1)==rd64(ip)) { matchl0=istart+el0-1; mLength=count_match(ip+8,matchl0+8,iend)+8; offset=(u32)(ip-matchl0);
                while(((ip>anchor)&(matchl0>istart))&&(ip[-1]==matchl0[-1])){ip--;matchl0--;mLength++;} el1 = 0; goto _found; }
            el1 = readL((u32)(ip1-istart));
            if (es0) { if ((rd32(istart+es0-1)&0xFF)==(rd32(ip)&0xFF)) fetchS++; }
            if (es0 && rd32(istart+es0-1)==rd32(ip)) { matchs0=istart+es0-1; goto _next_long; }
            if (ip1>=nextStep){step++;nextStep+=256;}
            ip=ip1; ip1+=step; el0=el1;
        } while (ip1<=ilimit);
        return;
_next_long:
        mLength=count_match(ip+4,matchs0+4,iend)+4; offset=(u32)(ip-matchs0);
        if ((el1>1)&&(rd64(istart+el1-1)==rd64(ip1))) { const u8* m1=istart+el1-1; u32 l1=count_match(ip1+8,m1+8,iend)+8; if (l1>mLength){ip=ip1;mLength=l1;offset=(u32)(ip-m1);matchs0=m1;} }
        while(((ip>anchor)&(matchs0>istart))&&(ip[-1]==matchs0[-1])){ip--;matchs0--;mLength++;}
_found:
        off2=off1; off1=offset; nSeq++;
        if (step<4) { u32 const p1=(u32)(ip1-istart); insertL(p1);
            // out-of-order: curr+2 inserted after ip1 (> curr+2) into the same bucket erases ip1's insertion
            if (p1 > curr+2 && hash8(istart+p1,HL)==hash8(istart+curr+2,HL)) insL[p1]=0; }
_stored:
        ip+=mLength; anchor=ip;
        if (ip<=ilimit){
            u32 const ins=curr+2; insertL(ins); insertL((u32)(ip-2-istart)); insertS(ins); insertS((u32)(ip-1-istart));
            while((ip<=ilimit)&&((off2>0)&(rd32(ip)==rd32(ip-off2)))){ u32 r=count_match(ip+4,ip+4-off2,iend)+4; u32 t=off2;off2=off1;off1=t; insertS((u32)(ip-istart)); insertL((u32)(ip-istart)); ip+=r; anchor=ip; nSeq++; }
        }
    }
}
int main(int argc,char**argv){ FILE*f=fopen(argv[1],"rb"); u32 fs=atoi(argv[2]); u8* buf=malloc(fs+64); u32 n=0; memset(buf,0,fs+64);
    while(fread(buf,1,fs,f)==fs){ frame(buf,fs); n++; }
    printf("frames %u  searched positions %llu (%.2f per byte)  sequences %llu\n", n,(unsigned long long)nSearch,(double)nSearch/((double)n*fs),(unsigned long long)nSeq);
    printf("long reads %llu: random reads by chain walk %llu (%.2f per read), empty %llu\n",(unsigned long long)nReadL,(unsigned long long)walkReadsL,(double)walkReadsL/nReadL,(unsigned long long)emptyL);
    printf("short reads %llu: random reads by chain walk %llu (%.2f per read), empty %llu\n",(unsigned long long)nReadS,(unsigned long long)walkReadsS,(double)walkReadsS/nReadS,(unsigned long long)emptyS);
    printf("walks > 8 steps: %llu   equivalence mismatches: %llu\n",(unsigned long long)longWalks,(unsigned long long)mismatch);
    printf("with a same-full-hash filter from pass A: long reads that need a walk %llu (%.1f%%), random reads %llu; short: %llu (%.1f%%), %llu -> %.2f random reads per searched position\n",
        (unsigned long long)needL,100.0*needL/nReadL,(unsigned long long)needReadsL,(unsigned long long)needS,100.0*needS/nReadS,(unsigned long long)needReadsS,(double)(needReadsL+needReadsS)/nSearch);
    printf("per searched position: now 2 table reads + 2 table writes = 4 random requests; prev-link: %.2f random reads, 0 random writes\n",(double)(walkReadsL+walkReadsS)/nSearch);
    return 0; }
