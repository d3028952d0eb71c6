# GPU box: the KNN refresh alone (scripts/knn_follow_only.py), standard library and variants.  usage: bash scripts/ab_knn.sh name1 name2 ...
echo -n "standard   "; python scripts/knn_follow_only.py | tail -1
for n in "$@"; do echo -n "$n   "; SOAR_HIP_LIB=soar_amd/_lib/variants/$n.so python scripts/knn_follow_only.py | tail -1; done
echo -n "standard   "; python scripts/knn_follow_only.py | tail -1
