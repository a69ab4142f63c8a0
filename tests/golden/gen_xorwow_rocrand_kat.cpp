// Generates tests/golden/xorwow_rocrand_kat.json from rocRAND's host-callable XORWOW engine
// (/opt/rocm/include/rocrand/rocrand_xorwow.h), an implementation of the same recurrence and
// the same 2^67-draw subsequence jump that is independent of this repository.  rocRAND
// differs from cuRAND only in the four seed-scrambling constants (printed into the file), so
// the oracle run WITH THOSE CONSTANTS must reproduce these draws exactly.
//   build+run:  hipcc -O1 -o /tmp/gen_kat gen_xorwow_rocrand_kat.cpp && /tmp/gen_kat > xorwow_rocrand_kat.json
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <cstdio>

int main() {
    const unsigned long long seeds[] = {12345ULL, 0ULL, 0xdeadbeefcafef00dULL};
    const unsigned long long subs[] = {0ULL, 1ULL, 2ULL, 7ULL, 1000ULL, 2073599ULL, 8294399ULL, (1ULL << 33) + 5ULL};
    printf("{\n \"source\": \"rocRAND xorwow_engine (ROCm 7.2) host path\",\n");
    printf(" \"seed_constants\": {\"xor0\": %u, \"xor1\": %u, \"mul0\": %u, \"mul1\": %u},\n", 0x2c7f967fU, 0xa03697cbU,
           1228688033U, 2073658381U);
    printf(" \"cases\": [\n");
    bool first = true;
    for (unsigned long long seed : seeds)
        for (unsigned long long sub : subs) {
            rocrand_device::xorwow_engine e(seed, sub, 0ULL);
            printf("%s  {\"seed\": %llu, \"subsequence\": %llu, \"draws\": [", first ? "" : ",\n", seed, sub);
            for (int i = 0; i < 8; ++i)
                printf("%s%u", i ? ", " : "", e.next());
            printf("]}");
            first = false;
        }
    printf("\n ]\n}\n");
    return 0;
}
