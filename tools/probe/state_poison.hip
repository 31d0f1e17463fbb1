// state_poison.hip -- investigation tool: kernels that change ONE kind of per-CU state between two launches
// of the kernel under test, so that a launch-to-launch difference can be pinned on that state:
//   poison_regs_lds(pattern): every workgroup writes `pattern` into all 256 VGPRs of each of its 8 waves (512 threads,
//                             2 waves per SIMD = the occupancy of the training kernels) and into all 160 KB of LDS;
//   icache_sweep():           ~300 KB of straight-line code (v_add on two registers): evicts the 64 KB instruction caches,
//                             touches no LDS and 2 VGPRs.
// Build: hipcc --offload-arch=gfx950 -shared -fPIC -o libstate_poison.so state_poison.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#define R8(b) "v_mov_b32 v" #b "0, %0\n v_mov_b32 v" #b "1, %0\n v_mov_b32 v" #b "2, %0\n v_mov_b32 v" #b "3, %0\n v_mov_b32 v" #b "4, %0\n" \
              "v_mov_b32 v" #b "5, %0\n v_mov_b32 v" #b "6, %0\n v_mov_b32 v" #b "7, %0\n v_mov_b32 v" #b "8, %0\n v_mov_b32 v" #b "9, %0\n"

__global__ void __launch_bounds__(512, 2) k_poison(uint32_t pattern, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) lds[i] = pattern;
    __syncthreads();
    uint32_t keep = lds[(threadIdx.x * 37) % (160 * 1024 / 4)];
    const uint32_t p = __builtin_amdgcn_readfirstlane(pattern);
    // registers v0 .. v255 (decades 0..24 + 250..255), pattern from an SGPR
    asm volatile(R8() R8(1) R8(2) R8(3) R8(4) R8(5) R8(6) R8(7) R8(8) R8(9) R8(10) R8(11) R8(12) R8(13) R8(14) R8(15) R8(16) R8(17)
                 R8(18) R8(19) R8(20) R8(21) R8(22) R8(23) R8(24)
                 "v_mov_b32 v250, %0\n v_mov_b32 v251, %0\n v_mov_b32 v252, %0\n v_mov_b32 v253, %0\n v_mov_b32 v254, %0\n v_mov_b32 v255, %0\n"
                 "s_nop 7\n"
                 :
                 : "s"(p)
                 : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
                   "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
                   "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",
                   "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",
                   "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90",
                   "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
                   "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",
                   "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136",
                   "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151",
                   "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166",
                   "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181",
                   "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196",
                   "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211",
                   "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226",
                   "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241",
                   "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    if (keep == 0xdeadbeefu && pattern != 0xdeadbeefu) sink[0] = keep;   // (keeps the LDS traffic alive)
}

__global__ void __launch_bounds__(512, 2) k_icache(float *sink, int never)
{
    float a = threadIdx.x, b = 1.0f;
    asm volatile(".rept 75000\n v_add_f32 %0, %0, %1\n .endr\n" : "+v"(a) : "v"(b));
    if (never) sink[threadIdx.x] = a;
}

extern "C" __attribute__((visibility("default"))) int poison_regs_lds(uint32_t pattern, uint32_t *sink, void *stream)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_poison, dim3(1024), dim3(512), 160 * 1024, (hipStream_t)stream, pattern, sink);
    return (int)hipGetLastError();
}
extern "C" __attribute__((visibility("default"))) int icache_sweep(float *sink, void *stream)
{
    hipLaunchKernelGGL(k_icache, dim3(512), dim3(512), 0, (hipStream_t)stream, sink, 0);
    return (int)hipGetLastError();
}
