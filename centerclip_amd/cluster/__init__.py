"""MI355X-native mirror of the reference's ``modules/cluster`` package (hot-path subset):
same function / module names, argument meaning and error behaviour, backed by the HIP
kernels in libcenterclip_hip.so."""
from .cluster_utils import pairwise_distance, KKZ_init  # noqa: F401
from .fast_kmeans import batch_fast_kmedoids, batch_fast_kmedoids_with_split  # noqa: F401
from .cluster import TokenClusterInter, get_cluster_inter  # noqa: F401
