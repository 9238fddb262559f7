# A/B of library sets on one box: tools/ab_libs.sh "B C" <command...>   (stereovision-slam_amd/lib_<X>/*.so copied over lib/
# before each run of the command; the first set is left in place)
L=stereovision-slam_amd
SETS=$1; shift
for v in $SETS; do
cp $L/lib_$v/*.so $L/lib/
echo "== lib_$v"
"$@"
done
set -- $SETS
cp $L/lib_$1/*.so $L/lib/
