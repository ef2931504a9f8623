// experiment: speculative chain walking convergence
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
static uint32_t hash(uint64_t x, uint64_t y, uint64_t seed){
  uint64_t h=(x*0x1E3779B1ull+y*0x05EBCA77ull+seed*0x42B2AE3Dull+0x165667B1ull)&0xFFFFFFFFull;
  h^=h>>15; h=(h*0x2C1B3C6Dull)&0xFFFFFFFFull; h^=h>>12; h=(h*0x297A2D39ull)&0xFFFFFFFFull; h^=h>>15; return (uint32_t)h;}
static int W,H; static uint8_t*img;
static int quant(int d){ if(d<=-21)return -4; if(d<=-7)return -3; if(d<=-3)return -2; if(d<0)return -1; if(d==0)return 0; if(d<3)return 1; if(d<7)return 2; if(d<21)return 3; return 4;}
typedef struct {uint8_t x,px; int8_t sg;} Ev;
typedef struct {int A,B,C,N;} St;
static inline int step(St*s, Ev e){
  int px=e.px+(e.sg<0?-s->C:s->C); if(px<0)px=0; if(px>255)px=255;
  int err=e.x-px; if(e.sg<0)err=-err; err=(int8_t)err;
  int k=0; while((s->N<<k)<s->A)k++;
  s->A+=err<0?-err:err; s->B+=err;
  if(s->N==64){s->A>>=1;s->B>>=1;s->N>>=1;} s->N++;
  if(s->B+s->N<=0){s->B+=s->N; if(s->B<=-s->N)s->B=-s->N+1; if(s->C>-128)s->C--;}
  else if(s->B>0){s->B-=s->N; if(s->B>0)s->B=0; if(s->C<127)s->C++;}
  return k;
}
int main(int argc,char**argv){
  const char*kind=argc>1?argv[1]:"gradient"; W=H=argc>2?atoi(argv[2]):4096;
  img=malloc((size_t)W*H);
  if(!strcmp(kind,"pgm")){ FILE*f=fopen(argv[3],"rb"); int mv; fscanf(f,"P5 %d %d %d",&W,&H,&mv); fgetc(f); img=malloc((size_t)W*H); fread(img,1,(size_t)W*H,f); fclose(f);} else
  for(int y=0;y<H;y++)for(int x=0;x<W;x++){ uint32_t h=hash(x,y,2); int v;
    if(!strcmp(kind,"noise")) v=h%256; else { int t=((3*x+2*y)>>4)%256; int tri=(t<128?t:255-t)*2; int span=!strcmp(kind,"hard")?32:3; v=tri+(int)(h%(2*span+1))-span; if(v<0)v=0; if(v>255)v=255;} img[(size_t)y*W+x]=v;}
  Ev**ch=calloc(365,sizeof(Ev*)); size_t*cnt=calloc(365,sizeof(size_t)),*cap=calloc(365,sizeof(size_t));
  uint8_t*prev=calloc(W+2,1),*cur=calloc(W+2,1); size_t runs=0;
  for(int y=0;y<H;y++){ memcpy(cur+1,img+(size_t)y*W,W); prev[W+1]=prev[W]; cur[0]=prev[1];
    for(int i=1;i<=W;){ int ra=cur[i-1],rb=prev[i],rc=prev[i-1],rd=prev[i+1];
      int q=81*quant(rd-rb)+9*quant(rb-rc)+quant(rc-ra);
      if(q==0){ runs++; while(i<=W&&cur[i]==ra)i++; if(i<=W)i++; continue;}
      int sg=q<0?-1:1; int c=q<0?-q:q; int mx=ra>rb?ra:rb,mn=ra<rb?ra:rb; int px=rc>=mx?mn:(rc<=mn?mx:ra+rb-rc);
      if(cnt[c]==cap[c]){cap[c]=cap[c]?cap[c]*2:1024; ch[c]=realloc(ch[c],cap[c]*sizeof(Ev));}
      ch[c][cnt[c]++]=(Ev){cur[i],px,sg}; i++; }
    uint8_t*t=prev;prev=cur;cur=t; }
  size_t total=0,maxc=0; int used=0; for(int c=1;c<365;c++){total+=cnt[c]; if(cnt[c]>maxc)maxc=cnt[c]; used+=cnt[c]>0;}
  printf("%s %dx%d regular=%zu runs=%zu used=%d maxchain=%zu\n",kind,W,H,total,runs,used,maxc);
  int Js[]={1024,4096,16384}; int Ws[]={64,128,256,512,1024,2048};
  for(int ji=0;ji<3;ji++)for(int wi=0;wi<6;wi++){ int J=Js[ji],Wm=Ws[wi]; size_t jobs=0,missBC=0,missA=0;
    for(int c=1;c<365;c++){ size_t n=cnt[c]; if(n<=(size_t)J)continue;
      // exact states at every job boundary
      St s={4,0,0,1}; size_t nb=n/J; St*ex=malloc((nb+1)*sizeof(St));
      for(size_t i=0;i<n;i++){ if(i%J==0&&i/J<=nb)ex[i/J]=s; step(&s,ch[c][i]); }
      for(size_t j=1;j*J<n;j++){ size_t b=j*J; size_t st0=b>(size_t)Wm?b-Wm:0; 
        // N at st0 exact
        int N; { size_t i=st0; N = i<64? (int)i+1 : 33+(int)((i-63-1)%32); /* after first halving at event idx 63 (N==64 before it) */ }
        // compute N by simulation to be safe
        { int n2=1; /* closed form check skipped */ (void)n2; }
        St g={4,0,0,N}; if(st0==0)g=(St){4,0,0,1};
        // better A guess: none
        for(size_t i=st0;i<b;i++)step(&g,ch[c][i]);
        jobs++; if(g.B!=ex[j].B||g.C!=ex[j].C)missBC++; if(g.A!=ex[j].A)missA++; if(g.N!=ex[j].N){printf("N mismatch %d %d at %zu\n",g.N,ex[j].N,st0);return 1;} }
      free(ex);} 
    printf("J=%5d W=%4d jobs=%zu missBC=%zu (%.3f%%) missA=%zu (%.3f%%)\n",J,Wm,jobs,missBC,100.0*missBC/(jobs?jobs:1),missA,100.0*missA/(jobs?jobs:1)); }
  return 0;}
