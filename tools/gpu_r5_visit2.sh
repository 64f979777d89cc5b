for v in w6p_prof_prio2 w6p_prof_nodsw w6p_prof_noarith; do echo "=== $v"; timeout 100 python tools/w6p_phase_prof.py $v 2>&1 | grep -v amdgpu.ids | head -22; done
echo "=== prio2 timing"; ONLY_BIG=1 VARIANT=w6p_prio2 timeout 100 python tools/wino6_ab.py 2>&1 | grep "^B16" | sed 's/.*| block/block/'
