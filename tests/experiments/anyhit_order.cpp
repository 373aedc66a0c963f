// Experiment behind the connect stage's visiting order (test infrastructure: links the CPU oracle, nothing ships from here).
// Counts quad steps and triangle tests of shadow rays on the 1M-triangle soup (or, with a second argument, the bundled scene)
// for seven orders of a step's surviving children: 0 nearest first (what the kernels do), 1 slot order (no sort), 2 farthest
// first, 3 longest overlap with the segment first, 4 largest box first, 5 leaves first, 6 overlap with leaves favoured.
//   g++ -O2 -std=c++17 -pthread -I oracle tests/experiments/anyhit_order.cpp -L oracle -loracle -Wl,-rpath,$PWD/oracle -o /tmp/anyhit_order
// Result (200,000 shadow rays): soup 41.3 / 42.1 / 41.0 / 37.9 / 42.7 / 41.3 / 37.8 steps per ray (94.7 % occluded; an
// occluded ray needs ~40 steps under every order, an unoccluded one 65.5); bundled scene 14.5 / 14.4 / 14.0 / 13.9 / 14.4 /
// 14.4 / 13.8.  No order changes the picture: the soup's triangles are small against their boxes, so finding any occluder
// costs about what finding the nearest one costs.  Both alternatives were then built into the kernels and the oracle and
// measured on the real frame (1M soup, 1080p, 64 spp; connect stage 61.3 ms with nearest first): slot order without a sort
// 67.3 ms (42.0 steps per shadow ray instead of 39.5), longest stretch first 63.0 ms (39.4 steps: the render's own shadow
// rays gain far less than the synthetic ones here, and the keys cost more arithmetic).  Nearest first stayed.
#include "oracle.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <random>
#include <thread>
#include <algorithm>
using namespace std; using namespace oracle;
static const float INF = __builtin_huge_valf();
struct E { bool leaf; int first, count, node; float key; };
struct Stats { uint64_t steps=0, tris=0, rays=0, occ=0, steps_occ=0, steps_free=0; };

// mode 0: near-first (current). 1: slot order (no sort). 2: far-first. 3: longest overlap first. 4: largest box area first. 5: leaves first then near
static bool walk(const Accelerator& acc, const Ray& ray, float tlimit, int mode, Stats& st) {
    const auto& nodes = acc.nodes; const auto& tr = acc.triangles;
    st.rays++;
    uint64_t steps0 = st.steps;
    if (!BoxIntersect(nodes[0].Bounds, ray, nullptr, nullptr)) return false;
    vector<E> stack;
    E cur{false,0,0,0,0};
    bool occluded=false;
    auto pop=[&](){ if(stack.empty()) return false; cur=stack.back(); stack.pop_back(); return true; };
    for(;;){
        if(cur.leaf){
            for(int j=cur.first;j<cur.first+cur.count;j++){ st.tris++; Intersection p; p.Distance=INF; p.Tri=-1; if(TriangleIntersect(tr[j],ray,p,j)&&p.Distance<tlimit){occluded=true;break;} }
            if(occluded) break;
            if(!pop()) break;
            continue;
        }
        st.steps++;
        int kids[4]; int nk=0;
        for(int ch=nodes[cur.node].LeftFirst; ch<=nodes[cur.node].LeftFirst+1; ch++){ if(nodes[ch].Count>0) kids[nk++]=ch; else {kids[nk++]=nodes[ch].LeftFirst; kids[nk++]=nodes[ch].LeftFirst+1;} }
        float keys[4]; int who[4]; int nh=0;
        for(int k=0;k<nk;k++){
            float tm, tx; bool h=BoxIntersect(nodes[kids[k]].Bounds,ray,&tm,&tx); h=h&&tm<tlimit; if(!h) continue;
            float key;
            const Box& b=nodes[kids[k]].Bounds;
            switch(mode){
                case 0: key=max(tm,0.f); break;
                case 1: key=(float)k; break;
                case 2: key=-tm; break;
                case 3: key=-(min(tx,tlimit)-max(tm,0.f)); break;
                case 4: { float dx=b.Max.x-b.Min.x,dy=b.Max.y-b.Min.y,dz=b.Max.z-b.Min.z; key=-(dx*dy+dy*dz+dx*dz);} break;
                case 5: key=(nodes[kids[k]].Count>0? -1e30f : max(tm,0.f)); break;
                case 6: { // density: triangles per overlap? leaves: count / volume ~ prefer small leaves? use overlap/diag
                    key = -(min(tx,tlimit)-max(tm,0.f)) * (nodes[kids[k]].Count>0 ? 4.f:1.f); } break;
                default: key=0;
            }
            keys[nh]=key; who[nh]=kids[k]; nh++;
        }
        for(int a=1;a<nh;a++) for(int b=a;b>0&&keys[b]<keys[b-1];b--){swap(keys[b],keys[b-1]);swap(who[b],who[b-1]);}
        auto ent=[&](int s){ const Node& n=nodes[who[s]]; E e; if(n.Count>0){e={true,n.LeftFirst,n.Count,0,0};} else e={false,0,0,who[s],0}; return e; };
        for(int a=nh-1;a>=1;a--) stack.push_back(ent(a));
        if(nh>0) cur=ent(0); else if(!pop()) break;
    }
    if(occluded){st.occ++; st.steps_occ+=st.steps-steps0;} else st.steps_free+=st.steps-steps0;
    return occluded;
}
int main(int argc,char**argv){
    int N = argc>1? atoi(argv[1]) : 200000;
    Scene s; bool bundled = argc>2;
    if(bundled){ s.meshSource="pathtracer_amd/assets/meshes.bin"; s.InitializeScene(11);} else s.GenerateSoup(1000000, 0x9E3779B9u);
    s.Build();
    LightTable lt = BuildLightTable(s);
    printf("built %zu nodes\n", s.accelerator.nodes.size());
    mt19937 rng(5); uniform_real_distribution<float> U(0,1); normal_distribution<float> G(0,1);
    vector<Ray> rays; vector<float> tmax;
    while((int)rays.size()<N){
        Vec3 o{U(rng)*100-50,U(rng)*100-50,U(rng)*100-50};
        if(bundled) o=Vec3{U(rng)*16-8,U(rng)*8,U(rng)*16-8}; Vec3 d{G(rng),G(rng),G(rng)}; d=Normalize(d);
        Ray r=MakeRay(o,d); Intersection i; i.Distance=INF; i.Tri=-1; s.accelerator.IntersectCanonical(r,i);
        if(i.Tri<0) continue;
        Vec3 p=o+d*i.Distance;
        // geometric normal of hit triangle, cosine test like the shader (light above)
        Vec3 y{U(rng)*60-30,60,U(rng)*60-30};
        if(bundled){ float r=U(rng)*lt.totalArea; int li=int(lower_bound(lt.cdf.begin(),lt.cdf.end(),r)-lt.cdf.begin()); if(li>=(int)lt.tris.size()) li=lt.tris.size()-1; const Triangle& L=s.accelerator.triangles[lt.tris[li]]; float su=sqrtf(U(rng)), b1=U(rng)*su, b0=1-su; y=L.V1.Position*b0+L.V2.Position*b1+L.V3.Position*(1-b0-b1); } Vec3 w=y-p; float dl=sqrtf(Dot(w,w)); Vec3 wn=w*(1.0f/dl);
        const Triangle& t=s.accelerator.triangles[i.Tri]; Vec3 n=t.V1.Normal; if(Dot(n,d)>0) n=n*-1.0f;   // facing normal (approx of shader's interpolated normal)
        if(Dot(wn,n)<=0) continue;
        rays.push_back(MakeRay(p+wn*1e-5f, wn)); tmax.push_back(dl*0.999f);
    }
    printf("%zu shadow rays\n", rays.size());
    for(int mode=0;mode<7;mode++){
        int T=8; vector<Stats> st(T); vector<thread> th;
        for(int t=0;t<T;t++) th.emplace_back([&,t]{ for(size_t i=t;i<rays.size();i+=T) walk(s.accelerator,rays[i],tmax[i],mode,st[t]); });
        for(auto&x:th)x.join();
        Stats a; for(auto&x:st){a.steps+=x.steps;a.tris+=x.tris;a.rays+=x.rays;a.occ+=x.occ;a.steps_occ+=x.steps_occ;a.steps_free+=x.steps_free;}
        printf("mode %d: steps/ray %.2f tris/ray %.2f occluded %.3f steps(occ) %.2f steps(free) %.2f\n",mode,(double)a.steps/a.rays,(double)a.tris/a.rays,(double)a.occ/a.rays,(double)a.steps_occ/max<uint64_t>(1,a.occ),(double)a.steps_free/max<uint64_t>(1,a.rays-a.occ));
    }
}
