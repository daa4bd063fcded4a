// Replay of the VALU arithmetic of the unsplit iiwa-7 forward-dynamics-gradient kernel (its fp instructions in program order with their
// own registers, everything else -- staging, LDS, stores, waits, address arithmetic -- removed), one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(64, 2) void replay(unsigned long long *cyc) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
#include "valu_replay_iiwa7_dFD.inc"
        ::: "memory", "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *d; CHECK(hipMalloc(&d, 8 * 8192));
    hipFuncAttributes a; CHECK(hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&replay)));
    printf("VALU replay of the unsplit iiwa-7 dFD kernel: 6998 arithmetic instructions, %d registers\n", a.numRegs);
    for (int blocks : {256, 1024, 2048, 4096}) {
        for (int r = 0; r < 20; r++) replay<<<blocks, 64>>>(d);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        auto p = [&](double q){ return (double)h[(size_t)(q * (blocks - 1))]; };
        double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
        printf("waves %5d (%.1f per SIMD): cycles per wave p0 %6.0f p10 %6.0f p50 %6.0f p90 %6.0f p100 %6.0f mean %6.0f | cycles/instr (median) %.2f | SIMD throughput vs one wave per SIMD: see mean\n",
               blocks, blocks / 1024.0, p(0), p(0.1), p(0.5), p(0.9), p(1), mean, p(0.5) / 6998);
    }
    return 0;
}
