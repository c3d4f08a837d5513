// DPP-fused strand pick (wide builds): compare-by-borrow, min and select read the previous lane's word through DPP
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <bool TIE_RC>
__global__ void k(uint32_t *o, const uint32_t *a)
{
    uint32_t ft = a[threadIdx.x], rt = a[threadIdx.x + 64], fl = a[threadIdx.x + 128], rl = a[threadIdx.x + 192], T, lo, tmp; uint64_t F;
    if (TIE_RC)
        asm("s_nop 1\n"
            "v_sub_co_u32_dpp %[tmp], vcc, %[ft], %[rt] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_min_u32_dpp %[T], %[ft], %[rt] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_cndmask_b32_dpp %[lo], %[rl], %[fl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "s_mov_b64 %[F], vcc\n" : [T]"=&v"(T), [lo]"=&v"(lo), [tmp]"=&v"(tmp), [F]"=s"(F) : [ft]"v"(ft), [rt]"v"(rt), [fl]"v"(fl), [rl]"v"(rl) : "vcc");
    else
        asm("s_mov_b64 vcc, -1\n"
            "v_subb_co_u32_dpp %[tmp], vcc, %[ft], %[rt], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_min_u32_dpp %[T], %[ft], %[rt] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "v_cndmask_b32_dpp %[lo], %[rl], %[fl], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
            "s_mov_b64 %[F], vcc\n" : [T]"=&v"(T), [lo]"=&v"(lo), [tmp]"=&v"(tmp), [F]"=s"(F) : [ft]"v"(ft), [rt]"v"(rt), [fl]"v"(fl), [rl]"v"(rl) : "vcc");
    o[threadIdx.x] = T; o[threadIdx.x + 64] = lo; o[threadIdx.x + 128] = (uint32_t)(F >> (threadIdx.x & 63)) & 1;
}
int main()
{
    uint32_t h[256], *d, *o, r[192];
    for (int i = 0; i < 256; i++) h[i] = (uint32_t)(i * 2654435761u) >> (i % 3);
    for (int i = 5; i < 60; i += 7) h[i + 64] = h[i - 1];   // ties
    h[9] = 0xFFFFFFFFu; h[10 + 64] = 0xFFFFFFFFu; h[20] = 0; h[21 + 64] = 0; h[30 + 64] = 0xFFFFFFFFu; h[40 + 64] = 0;
    (void)hipMalloc(&d, 1024); (void)hipMalloc(&o, 768); (void)hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    int total = 0;
    for (int tie = 0; tie < 2; tie++) {
        if (tie) k<true><<<1, 64>>>(o, d); else k<false><<<1, 64>>>(o, d);
        (void)hipMemcpy(r, o, 768, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) {
            uint32_t ft = l ? h[l - 1] : 0, rt = h[l + 64], fl = h[l + 128], rl = l ? h[l + 191] : 0;
            bool fwd = tie ? ft < rt : ft <= rt; uint32_t T = ft < rt ? ft : rt, lo = fwd ? fl : rl;
            if (r[l] != T || r[l + 64] != lo || r[l + 128] != (uint32_t)fwd) { bad++; if (bad < 5) printf("lane %d: T %x/%x lo %x/%x F %u/%u\n", l, r[l], T, r[l+64], lo, r[l+128], (unsigned)fwd); }
        }
        printf("dpp pick tie_rc=%d: %d bad lanes\n", tie, bad); total += bad;
    }
    return total != 0;
}
