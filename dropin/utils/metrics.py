from demo2_amd.utils.metrics import *  # noqa: F401,F403  (R1_mAP_eval, R1_mAP, eval_func, eval_func_msrv, euclidean_distance)
from demo2_amd.utils.metrics import R1_mAP, R1_mAP_eval, eval_func, eval_func_msrv, euclidean_distance  # noqa: F401
