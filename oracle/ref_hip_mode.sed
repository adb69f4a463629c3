# The edit a maintainer of mjoppich/tsxCount makes to src/mains/main.cpp to add --mode=HIP
# (INTEGRATION.md section 2), as a sed script: oracle/Makefile pipes the reference's main.cpp
# through it straight into the compiler (no patched copy is written anywhere).
# 1. the binding
/#include <tsxcount\/TSXHashMapOMPPerfCount.h>/a\
#include <TSXHashMapHIP.h>
# 2. the mode: enum (main.cpp:41), its name (:42), strToMode (:53-81)
s/EXPERIMENTAL, OMP_COUNT };/EXPERIMENTAL, OMP_COUNT, HIP };/
s/"EXPERIMENTAL", "OMP_COUNT" };/"EXPERIMENTAL", "OMP_COUNT", "HIP" };/
/^    return tsx_mode::TRANSACTIONAL;/i\
    if (argStr == "HIP") { return tsx_mode::HIP; }
# 3. the map (main.cpp:429-475)
/^        case SERIAL:/i\
        case HIP:\
            std::cerr << "Creating TSXHashMap HIP" << std::endl;\
            pMap = new TSXHashMapHIP(arguments.l, arguments.storagebits, arguments.k, arguments.threads);\
            break;
# 4. getKmerCount() and getKmerStartsRef() are not virtual: the HIP map fills the base class's
#    bitmap once, after the counting loop (main.cpp:222)
/Added a total of/i\
    if (TSXHashMapHIP* pHip = dynamic_cast<TSXHashMapHIP*>(pMap)) { pHip->finish(); }
