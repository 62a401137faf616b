// developer check: the restated glibc pow (as in k_adv.hip gpow15) against libm pow.  gcc -O2 -mfma -ffp-contract=off tools/check_glibc_pow_clone.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../extpom_amd/csrc/glibc_pow_tables.h"
static const double A[7] = GPOW_A;
static const double LT[128][3] = GPOW_LOGTAB;
static const double C[4] = GEXP_C;
static const uint64_t ET[256] = GEXP_TAB;
static inline uint64_t asu(double x){uint64_t u;memcpy(&u,&x,8);return u;}
static inline double asd(uint64_t u){double x;memcpy(&x,&u,8);return x;}
#define FMA __builtin_fma
// x>0 normal, result normal (the fast path of __ieee754_pow_fma, instruction for instruction)
double gpow(double x, double y){
  uint64_t ix=asu(x);
  uint64_t tmp=ix-0x3fe6955500000000ULL;
  int i=(tmp>>45)&127; int64_t k=(int64_t)tmp>>52;
  uint64_t iz=ix-(tmp&0xfff0000000000000ULL);
  double z=asd(iz), kd=(double)k;
  double invc=LT[i][0], logc=LT[i][1], logctail=LT[i][2];
  double t1=FMA(kd,GPOW_LN2HI,logc);
  double r=FMA(z,invc,-1.0);
  double ar=r*A[0];
  double lo1=FMA(kd,GPOW_LN2LO,logctail);
  double q1=FMA(r,A[2],A[1]);
  double q2=FMA(r,A[4],A[3]);
  double t2=r+t1;
  double ar2=r*ar;
  double d1=t1-t2;
  double ar3=r*ar2;
  double lo3=FMA(ar,r,-ar2);
  double lo2=d1+r;
  double q3=FMA(r,A[6],A[5]);
  double hi=t2+ar2;
  double d2=t2-hi;
  double q4=FMA(q3,ar2,q2);
  double lo4=d2+ar2;
  double q5=FMA(ar2,q4,q1);
  double s=lo1+lo2; s=s+lo3; s=s+lo4;
  double lo=FMA(ar3,q5,s);
  double yl=hi+lo;
  double tl=(hi-yl)+lo;
  double ehi=y*yl;
  double e2=FMA(yl,y,-ehi);
  double elo=FMA(y,tl,e2);
  // exp_inline(ehi, elo)
  double kd2=FMA(ehi,GEXP_INVLN2N,GEXP_SHIFT);
  uint64_t ki=asu(kd2);
  kd2=kd2-GEXP_SHIFT;
  double rr=FMA(kd2,GEXP_NEGLN2HIN,ehi);
  rr=FMA(kd2,GEXP_NEGLN2LON,rr);
  unsigned idx=2*(ki&127);
  uint64_t sbits=ET[idx+1]+(ki<<45);
  rr=elo+rr;
  double p1=FMA(rr,C[1],C[0]);
  double s1=rr+asd(ET[idx]);
  double r2=rr*rr;
  double p2=FMA(rr,C[3],C[2]);
  double s2=FMA(p1,r2,s1);
  double r4=r2*r2;
  double tm=FMA(p2,r4,s2);
  double sc=asd(sbits);
  return FMA(tm,sc,sc);
}
int main(){
  srand48(3); long bad=0,n=20000000, notcr=0;
  for(long t=0;t<n;t++){
    double x = (t%3==0)? 30.+10.*drand48() : ((t%3==1)? drand48()*50. : exp(40*(drand48()-0.5)));
    double a=pow(x,1.5), b=gpow(x,1.5);
    if(a!=b){ if(bad<5) printf("x=%a glibc=%a clone=%a\n",x,a,b); bad++; }
  }
  printf("x**1.5: %ld mismatches of %ld\n",bad,n);
  bad=0;
  for(long t=0;t<2000000;t++){ double x=exp(20*(drand48()-0.5)), y=4*(drand48()-0.5); double a=pow(x,y), b=gpow(x,y); if(a!=b) bad++; }
  printf("general x**y: %ld mismatches of 2000000\n",bad);
  return 0;}
