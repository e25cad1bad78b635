// thp_probe.c -- what first touch and release of the host's big arrays cost on this box, with 4 KiB pages and with transparent huge
// pages (gcc -O2 -pthread -o /tmp/thp_probe tools/thp_probe.c; /tmp/thp_probe <GiB> <threads>)
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include <sys/mman.h>
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}
typedef struct{char*p;size_t n;}job;
static void*touch(void*v){job*j=v;for(size_t i=0;i<j->n;i+=4096)j->p[i]=1;return 0;}
static void run(const char*name,size_t bytes,int nt,int huge){
  double t0=now();
  char*p=mmap(0,bytes,PROT_READ|PROT_WRITE,MAP_PRIVATE|MAP_ANONYMOUS,-1,0);
  if(p==MAP_FAILED){perror("mmap");return;}
  if(huge) madvise(p,bytes,MADV_HUGEPAGE);
  pthread_t th[64];job jb[64];
  double t1=now();
  for(int i=0;i<nt;i++){jb[i].p=p+bytes/nt*i;jb[i].n=bytes/nt;pthread_create(&th[i],0,touch,&jb[i]);}
  for(int i=0;i<nt;i++)pthread_join(th[i],0);
  double t2=now();
  munmap(p,bytes);
  double t3=now();
  printf("%-28s map %.1f ms, first touch (%d threads) %.1f ms, unmap %.1f ms\n",name,t1-t0,nt,t2-t1,t3-t2);
}
int main(int argc,char**argv){
  size_t gib=argc>1?strtoull(argv[1],0,10):8; int nt=argc>2?atoi(argv[2]):16;
  FILE*f=fopen("/sys/kernel/mm/transparent_hugepage/enabled","r");char b[256]="?";if(f){if(!fgets(b,sizeof b,f))b[0]=0;fclose(f);}printf("transparent_hugepage/enabled: %s",b);
  f=fopen("/sys/kernel/mm/transparent_hugepage/defrag","r");if(f){if(fgets(b,sizeof b,f))printf("transparent_hugepage/defrag: %s",b);fclose(f);}
  run("4 KiB pages",gib<<30,nt,0); run("MADV_HUGEPAGE",gib<<30,nt,1); run("4 KiB pages again",gib<<30,nt,0); run("MADV_HUGEPAGE again",gib<<30,nt,1);
  return 0;
}
