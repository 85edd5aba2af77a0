#!/usr/bin/env python3
"""Makes an INSTRUMENTED copy of pion_amd/csrc (timing experiment, not a product build):
    python3 profiles/tools/instrument_stage_timing.py /tmp/exp_timing
    SRC=/tmp/exp_timing profiles/tools/build_variant.sh timing
    PION_DBG=1 PION_GPU_LIB=$PWD/ab/timing/libpion_gpu.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-build
k_stage_rows2 then reads the shader clock (s_memtime) at the start of every task, after an s_waitcnt vmcnt(0) placed
behind the task's loads, and at the task's end, and again around the cell update; per launch the launcher prints the
per-wavefront sums: cycles waiting for loads / running the body per task kind, the update, the whole wavefront, and a
histogram of hardware wave slots.  The forced full wait replaces the compiler's partial waits in the second-order
instance (the first-order instance has no stamp between loads and body: everything is "comp")."""
import shutil, sys, os

dst = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "..", "pion_amd", "csrc")
if os.path.exists(dst):
    shutil.rmtree(dst)
shutil.copytree(src, dst, ignore=shutil.ignore_patterns("build", "*.so", "*.o"))
p = os.path.join(dst, "stage_rows2.h")
s = open(p).read()


def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)


rep("PDEV double ldu_once(const char *ubase, const unsigned off)",
    """static __device__ unsigned long long g_dbgt[32];
PDEV unsigned long long stamp(bool drain)
{
  unsigned long long t;
  if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  else asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
PDEV double ldu_once(const char *ubase, const unsigned off)""")
rep("  bool pf_valid = false;",
    "  unsigned long long tw[3] = {0,0,0}, tcmp[3] = {0,0,0}, tup = 0, tk0 = stamp(false);\n  bool pf_valid = false;")
rep("        constexpr int TC = decltype(tc)::value;\n",
    "        constexpr int TC = decltype(tc)::value;\n        const unsigned long long tsA = stamp(false);\n        unsigned long long tsB = tsA;\n")
rep("""              qp[v] = ldu(St + v * ncb + 8, off);
            }
""", """              qp[v] = ldu(St + v * ncb + 8, off);
            }
            tsB = stamp(true);
""")
rep("            hslope3<NV>(A, B, C, dx, thr, sB);", "            tsB = stamp(true);\n            hslope3<NV>(A, B, C, dx, thr, sB);")
rep("              cyl_slope3<NV>(A, B, C, cA, cB, cC, true, sB);", "              tsB = stamp(true);\n              cyl_slope3<NV>(A, B, C, cA, cB, cC, true, sB);")
rep("            hslope3<NV>(zq0, qp1, qp2, dx, thr, sn);", "            tsB = stamp(true);\n            hslope3<NV>(zq0, qp1, qp2, dx, thr, sn);")
rep("""          for (int v = 0; v < NV; v++) ZS2(r, NZ - NV + v) = f[v];
        }
      };""", """          for (int v = 0; v < NV; v++) ZS2(r, NZ - NV + v) = f[v];
        }
        const unsigned long long tsC = stamp(false);
        const int ti = (t == 0) ? 0 : (t == 2 ? 1 : 2);
        tw[ti] += tsB - tsA;
        tcmp[ti] += tsC - tsB;
      };""")
rep("      if (!prime && writer && row_ok) {\n        const unsigned off = pin_v(off_r), offb = pin_v(offb_r);",
    "      const unsigned long long sU = stamp(false);\n      if (!prime && writer && row_ok) {\n        const unsigned off = pin_v(off_r), offb = pin_v(offb_r);")
rep("""            tmp = (t < tmp) ? t : tmp;
          }
        }
      }
    }
  }
#undef ZS2
""", """            tmp = (t < tmp) ? t : tmp;
          }
        }
      }
      tup += stamp(true) - sU;
    }
  }
#undef ZS2
  {
    const unsigned long long te = stamp(false);
    if (lane == 0) {
      for (int i = 0; i < 3; i++) { atomicAdd(&g_dbgt[i], tw[i]); atomicAdd(&g_dbgt[3 + i], tcmp[i]); }
      atomicAdd(&g_dbgt[6], tup);
      atomicAdd(&g_dbgt[7], te - tk0);
      atomicAdd(&g_dbgt[8], 1ull);
      const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 15u;
      atomicAdd(&g_dbgt[16 + slot], 1ull);
      atomicAdd(&g_dbgt[9 + (slot & 1)], te - tk0);
    }
  }
""")
rep("  hipLaunchKernelGGL(KERNEL, dim3((unsigned)nblocks), dim3(256), shmem, s, a);\n  return (int)hipGetLastError();",
    """  hipLaunchKernelGGL(KERNEL, dim3((unsigned)nblocks), dim3(256), shmem, s, a);
  if (getenv("PION_DBG")) {
    hipStreamSynchronize(s);
    unsigned long long h[32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbgt), sizeof(h));
    const double w = (double)h[8];
    fprintf(stderr, "DBG ooa=%d waves=%.0f total/wave=%.0f  wait x/y/z=%.0f %.0f %.0f  comp x/y/z=%.0f %.0f %.0f  upd=%.0f  even/odd-slot cycles %.3g %.3g slots", a.space_ooa, w, h[7] / w,
            h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, (double)h[9], (double)h[10]);
    for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", h[16 + i]);
    fprintf(stderr, "\\n");
    unsigned long long z[32] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbgt), z, sizeof(z));
  }
  return (int)hipGetLastError();""")
open(p, "w").write(s)
print("instrumented copy in", dst)
