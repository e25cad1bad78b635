// replay_fixed_point.c -- the in-place rehash of encap_kmerset (newhash.c:359-406) as a FIXED POINT of insertion times, checked against
// the sequential emulation (host only; gcc -O2 -o /tmp/fp tools/replay_fixed_point.c -lm; /tmp/fp <keys> [initial size] [seed]).
// An old entry at slot q is inserted at time (q, 0) unless its slot is taken earlier by an entry inserted at time t: then it gives way and
// is carried on at once, time t + 1.  Starting from (q, 0) for everybody, lay the entries out first come first served by time, read the
// evictions off the layout, repeat (all times of a round from the layout of the round before: Jacobi, not Gauss-Seidel): times only
// fall, never below the true ones, and the only fixed point is the sequential run.  10-15 rounds per growth on millions of keys.  Every
// round is a priority insertion (parallel, order independent): this is what sdt_gpu_layout_on_device runs (csrc/sdt_graph_kernels.cuh).
// Third table C: the same fixed point with INCREMENTAL rounds, as the device runs it since round 5.  The set of occupied slots of a
// linear-probing table does not depend on the order of insertion, so the clusters (maximal runs of occupied slots) of the new table
// are the same in every round, and the word at slot i depends on the entries with a home at or before i only: an entry of home h
// whose time changed can only re-arrange the slots from h to the end of its cluster.  After the first full round, a round takes those
// stretches out (a walker stops at a slot another one emptied: that one goes on), re-inserts their entries with the new times, and
// re-evaluates only the old slots that lie inside them (the evaluation is idempotent: f(w, f(w, t)) = f(w, t)).  About 0.5 of one
// full round in all.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static int prime_kh(uint64_t num){if(num<4)return 1;if(num%2==0)return 0;uint64_t lim=(uint64_t)sqrt((float)num);for(uint64_t i=3;i<lim;i+=2)if(num%i==0)return 0;return 1;}
static uint64_t next_prime_kh(uint64_t n){if(n%2==0)n++;while(!prime_kh(n))n+=2;return n;}
static uint64_t next_size(uint64_t size,double lf,uint64_t count){uint64_t n=size;do{n=n<0xFFFFFFFu?n<<1:n+0xFFFFFFu;n=next_prime_kh(n);}while(n*lf<(double)(count+1));return n;}
typedef struct { uint64_t t; uint32_t id; } ent;
static int cmp(const void*a,const void*b){const ent*x=a,*y=b;return x->t<y->t?-1:x->t>y->t;}
int main(int argc,char**argv){
  uint64_t m=argc>1?strtoull(argv[1],0,10):200000; uint64_t init=argc>2?strtoull(argv[2],0,10):next_prime_kh(1024); uint64_t seed=argc>3?strtoull(argv[3],0,10):1;
  uint64_t *keys=malloc(m*8); uint64_t x=88172645463325252ULL^(seed*0x9E3779B97F4A7C15ULL);
  for(uint64_t i=0;i<m;i++){x^=x<<13;x^=x>>7;x^=x<<17;keys[i]=x>>2;}
  // two tables: A = sequential reference (in place), B = fixed point
  uint64_t size=init,count=0,max=(uint64_t)(size*0.77f); double lf=(double)0.77f;
  uint64_t cap=1; { uint64_t s=init,mx=max; while(m>mx){s=next_size(s,lf,mx);mx=(uint64_t)(s*lf);} cap=s; }
  uint32_t *A=calloc(cap,4), *B=calloc(cap,4), *NB=calloc(cap,4), *C=calloc(cap,4); uint64_t *CW=calloc(cap,8), *ct=malloc(cap*8), *ch=malloc(cap*8), *LA=malloc(cap*8), *LB=malloc(cap*8), *LW=malloc(cap*8); uint8_t *dirty=calloc(cap,1); uint64_t full_ins=0, inc_ins=0, inc_eval=0; int maxr2=0; uint8_t *flag=calloc(cap,1);
  uint64_t *tm=malloc(m*8), *tn=malloc(m*8); ent *lst=malloc(m*sizeof(ent));
  int maxrounds=0; uint64_t grows=0;
  for(uint64_t i=0;i<m;i++){
    if(count+1>max){
      uint64_t old=size,n=next_size(size,lf,count); grows++;
      // --- A: sequential in place: flag 1 = old unmoved, 2 = new placed
      for(uint64_t j=0;j<old;j++) flag[j]=A[j]?1:0; for(uint64_t j=old;j<n;j++){flag[j]=0;A[j]=0;}
      for(uint64_t j=0;j<old;j++){ if(flag[j]!=1)continue; uint32_t carry=A[j]; flag[j]=0; A[j]=0;
        for(;;){ uint64_t h=keys[carry-1]%n; while(flag[h]==2)h=h+1==n?0:h+1;
          if(h<old&&flag[h]==1){uint32_t ev=A[h];A[h]=carry;flag[h]=2;carry=ev;continue;}
          A[h]=carry;flag[h]=2;break; } }
      // --- B: fixed point
      uint64_t cnt=0; for(uint64_t q=0;q<old;q++) if(B[q]){ tm[B[q]-1]=q<<20; }
      int rounds=0;
      for(;;){ rounds++;
        cnt=0; for(uint64_t q=0;q<old;q++) if(B[q]){ lst[cnt].t=tm[B[q]-1]; lst[cnt].id=B[q]; cnt++; }
        qsort(lst,cnt,sizeof(ent),cmp);
        memset(NB,0,n*4);
        for(uint64_t k=0;k<cnt;k++){ uint64_t h=keys[lst[k].id-1]%n; while(NB[h])h=h+1==n?0:h+1; NB[h]=lst[k].id; }
        int changed=0;
        for(uint64_t q=0;q<old;q++) if(B[q]){ uint32_t y=B[q], xo=NB[q]; uint64_t nt=q<<20;
          if(xo==y) nt=tm[y-1]; else if(xo&&tm[xo-1]<(q<<20)) nt=tm[xo-1]+1;
          tn[y-1]=nt; if(nt!=tm[y-1]){ if(nt>tm[y-1]){printf("time went UP: round %d y=%u q=%llu old t=(%llu,%llu) new t=(%llu,%llu) occupant %u t=(%llu,%llu)\n",rounds,y,(unsigned long long)q,(unsigned long long)(tm[y-1]>>20),(unsigned long long)(tm[y-1]&0xFFFFF),(unsigned long long)(nt>>20),(unsigned long long)(nt&0xFFFFF),xo,xo?(unsigned long long)(tm[xo-1]>>20):0ULL,xo?(unsigned long long)(tm[xo-1]&0xFFFFF):0ULL);return 1;} changed=1; } }
        for(uint64_t q=0;q<old;q++) if(B[q]) tm[B[q]-1]=tn[B[q]-1];
        if(!changed)break;
        if(rounds>200){printf("no convergence\n");return 1;}
      }
      if(rounds>maxrounds)maxrounds=rounds;
      memcpy(B,NB,n*4);
      if(memcmp(A,B,n*4)){ printf("MISMATCH after growth to %llu at count %llu (rounds %d)\n",(unsigned long long)n,(unsigned long long)count,rounds); return 1; }

      // --- C: incremental rounds (what the device runs): table words CW[slot] = time << 32 | q + 1 over the NEW geometry, times ct[q], homes ch[q]
      { const uint64_t D=6; memset(CW,0,n*8); uint64_t nA=0,nB=0,nW=0; int r2=0;
        #define INS(q_) do{ uint64_t w_=(ct[q_]<<32)|((q_)+1), h_=ch[q_]; for(;;){ uint64_t c_=CW[h_]; if(!c_){CW[h_]=w_;break;} if(c_>w_){CW[h_]=w_;w_=c_;} h_=h_+1==n?0:h_+1; } }while(0)
        #define EVAL(q_) do{ uint64_t w_=CW[q_], x_=w_&0xFFFFFFFFu, tx_=w_>>32, mine_=ct[q_], scan_=(uint64_t)(q_)<<D, nt_=scan_; if(x_==(q_)+1)nt_=mine_; else if(w_&&tx_<scan_)nt_=tx_+1; \
            if(nt_!=mine_){ ct[q_]=nt_; LB[nB++]=ch[q_]; } }while(0)
        for(uint64_t q=0;q<old;q++) if(C[q]){ ct[q]=q<<D; ch[q]=keys[C[q]-1]%n; }
        for(uint64_t q=0;q<old;q++) if(C[q]){ INS(q); full_ins++; }
        for(uint64_t q=0;q<old;q++) if(C[q]) EVAL(q);
        while(nB){ r2++; memcpy(LA,LB,nB*8); nA=nB; nB=0; nW=0;
          for(uint64_t k=0;k<nA;k++){ uint64_t i=LA[k],len=0; while(CW[i]){ LW[nW++]=(CW[i]&0xFFFFFFFFu)-1; CW[i]=0; i=i+1==n?0:i+1; len++; } LA[k]|=len<<40; }
          for(uint64_t k=0;k<nW;k++){ INS(LW[k]); inc_ins++; }
          for(uint64_t k=0;k<nA;k++){ uint64_t i=LA[k]&0xFFFFFFFFFFULL,len=LA[k]>>40; for(uint64_t j=0;j<len;j++){ if(i<old&&C[i]){ EVAL(i); inc_eval++; } i=i+1==n?0:i+1; } }
          if(r2>300){printf("incremental: no convergence\n");return 1;} }
        if(r2>maxr2)maxr2=r2;
        for(uint64_t j=0;j<n;j++) NB[j]=CW[j]?C[(CW[j]&0xFFFFFFFFu)-1]:0;
        memcpy(C,NB,n*4);
        if(memcmp(A,C,n*4)){ printf("INCREMENTAL MISMATCH after growth to %llu (rounds %d)\n",(unsigned long long)n,r2); return 1; } }
      size=n; max=(uint64_t)(n*lf);
    }
    uint64_t h=keys[i]%size; while(A[h])h=h+1==size?0:h+1; A[h]=(uint32_t)(i+1);
    h=keys[i]%size; while(B[h])h=h+1==size?0:h+1; B[h]=(uint32_t)(i+1);
    h=keys[i]%size; while(C[h])h=h+1==size?0:h+1; C[h]=(uint32_t)(i+1);
    count++;
  }
  printf("m=%llu init=%llu: %llu growths, identical; most rounds for one growth: %d; incremental: %d rounds at most, %llu full + %llu incremental insertions, %llu incremental evaluations\n",(unsigned long long)m,(unsigned long long)init,(unsigned long long)grows,maxrounds,maxr2,(unsigned long long)full_ins,(unsigned long long)inc_ins,(unsigned long long)inc_eval);
  return 0;
}
