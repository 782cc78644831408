/* Sanitizer harness for the oracle (CPU only; GPU sanitizers are not available on the pool): built with
 * -fsanitize=address,undefined by tests/test_sanitizer_cpu.py and run on noisy input through every entry point. */
#include "orb_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void){
  int W=322,H=241; unsigned char* img=malloc(W*H); unsigned s=12345;
  for(int i=0;i<W*H;i++){ s=s*1664525u+1013904223u; img[i]=(unsigned char)((s>>24) & 0xFF); if((i/ W/16 + i%W/16)&1) img[i]=img[i]/4+90; }
  oro_extractor e; oro_extractor_init(&e,500,1.2f,8,20,7);
  int cap=4096,n=0; oro_keypoint* k=malloc(sizeof(oro_keypoint)*cap); unsigned char* d=malloc(cap*32);
  int rc=oro_extract(&e,img,W,H,W,k,d,cap,&n,NULL,NULL); printf("rc=%d n=%d\n",rc,n);
  int32_t *bi=malloc(4*n),*bd=malloc(4*n),*sd=malloc(4*n); oro_best2(d,n,d,n,NULL,NULL,bi,bd,sd);
  int *items=malloc(4*n); oro_grid g; oro_grid_build(&g,k,n,0,(float)W,0,(float)H,items);
  int32_t out[4096]; int c=oro_features_in_area(&g,k,100.f,100.f,40.f,-1,-1,out,4096); printf("area %d\n",c);
  float *x=malloc(4*n),*y=malloc(4*n),*r=malloc(4*n); int32_t *mn=malloc(4*n),*mx=malloc(4*n);
  for(int i=0;i<n;i++){x[i]=k[i].x;y[i]=k[i].y;r[i]=20.f;mn[i]=-1;mx[i]=-1;}
  oro_search_area_best2(&g,k,d,NULL,d,x,y,r,mn,mx,n,bi,bd,sd);
  /* stereo on the same image as both views */
  uint8_t* lv[16]; int lw[16],lh[16]; for(int l=0;l<8;l++){oro_level_size(&e,W,H,l,&lw[l],&lh[l]); lv[l]=malloc(lw[l]*lh[l]);}
  oro_compute_pyramid(&e,img,W,H,W,lv);
  float *u=malloc(4*n),*dp=malloc(4*n); oro_stereo_matches(&e,k,d,n,k,d,n,lv,lv,lw,lh,0.08f,40.f,u,dp);
  int m=0; for(int i=0;i<n;i++) m+=u[i]>=0; printf("stereo matched %d\n",m);
  return 0; }
