"""Hand-computed known-answer vectors for the validation metrics.

torchmetrics — the library behind the reference's engines/metrics.py:125-159 — is absent from /root/reference and from this image,
and the reference holds no metric fixtures: these cases are worked out BY HAND from the published definitions (arithmetic in the
comments), so that the CPU oracle (tests/test_metrics_cpu.py) and the device kernels (tests/test_metrics_gpu.py) are pinned to
numbers neither of them produced.  Order of every expectation: Acc, AUC, Precision, Recall, F1, CK, Acc_micro (oracle KEYS).
"""
import numpy as np

# ---- case A: multiclass, C = 3, probability inputs (rows sum to 1: no softmax), a tie inside one AUROC and a class that never
#      occurs as a target (class 2) but is predicted once
A_probs = np.array([[0.7, 0.2, 0.1],      # s0  label 0 -> pred 0  ok
                    [0.4, 0.5, 0.1],      # s1  label 0 -> pred 1  wrong
                    [0.2, 0.6, 0.2],      # s2  label 1 -> pred 1  ok
                    [0.1, 0.3, 0.6],      # s3  label 1 -> pred 2  wrong
                    [0.3, 0.6, 0.1],      # s4  label 1 -> pred 1  ok
                    [0.6, 0.3, 0.1]],     # s5  label 0 -> pred 0  ok
                   dtype=np.float32)
A_labels = np.array([0, 0, 1, 1, 1, 0])
# confusion [target, pred]: [[2,1,0],[0,2,1],[0,0,0]]
#   class 0: tp 2 fp 0 fn 1 -> P 1, R 2/3, F1 4/5;  class 1: tp 2 fp 1 fn 1 -> P 2/3, R 2/3, F1 4/6;
#   class 2: tp 0 fp 1 fn 0 -> tp+fp+fn = 1 > 0: it takes part in the macro averages with P 0, R 0 (0/0 -> 0), F1 0
#   macro P = (1 + 2/3 + 0)/3 = 5/9, macro R = (2/3 + 2/3 + 0)/3 = 4/9 (= macro Acc), macro F1 = (4/5 + 2/3 + 0)/3 = 22/45
#   micro Acc = 4/6;  kappa: po = 4/6, pe = (3*2 + 3*3 + 0*1)/36 = 15/36 -> (2/3 - 5/12)/(7/12) = 3/7
#   AUROC one-vs-rest: class 0: positives {.7,.4,.6} all above negatives {.2,.1,.3} -> 1;
#     class 1: positives {.6,.3,.6} vs negatives {.2,.5,.3}: 3 + (1 + 0 + 0.5) + 3 = 7.5 of 9 -> 5/6; class 2 has no positives: skipped
#   macro AUC = (1 + 5/6)/2 = 11/12
A_expect = np.array([4 / 9, 11 / 12, 5 / 9, 4 / 9, 22 / 45, 3 / 7, 4 / 6])

# ---- case B: the binary task (--bin_metric, C = 2) on logits[:, 1]; scores outside [0,1] -> sigmoid, hard label = prob > 0.5
B_logits = np.stack([np.zeros(8), np.array([2.0, -1.0, 0.0, 2.0, -1.0, 0.5, 0.0, 3.0])], 1).astype(np.float32)
B_labels = np.array([1, 0, 1, 0, 0, 1, 0, 1])
# sigmoid > 0.5  <=>  logit > 0: pred = [1,0,0,1,0,1,0,1]: tp 3 (s0,s5,s7), fp 1 (s3), fn 1 (s2: sigmoid(0) = 0.5 is NOT > 0.5), tn 3
#   Acc 6/8, P 3/4, R 3/4, F1 6/8, kappa: po = 3/4, pe = (4*4 + 4*4)/64 = 1/2 -> 1/2
#   AUROC: positives {2, 0, .5, 3}, negatives {-1, 2, -1, 0}: 2 -> 3 + tie .5; 0 -> 2 + tie .5; .5 -> 3; 3 -> 4: 13/16
B_expect = np.array([0.75, 13 / 16, 0.75, 0.75, 0.75, 0.5, 0.75])

# ---- case C: multiclass C = 2 on raw logits (softmax), every prediction right except one, a perfect-separation AUROC
C_logits = np.array([[3.0, 0.0], [2.0, 1.0], [0.0, 1.0], [0.5, 2.5], [1.5, 1.0]], dtype=np.float32)
C_labels = np.array([0, 0, 1, 1, 1])
# softmax p1 = sigmoid(l1 - l0) = sigmoid(-3, -1, 1, 2, -0.5): pred = [0,0,1,1,0]; confusion [[2,0],[1,2]]
#   class 0: tp 2 fp 1 fn 0 -> P 2/3 R 1 F1 4/5;  class 1: tp 2 fp 0 fn 1 -> P 1 R 2/3 F1 4/5
#   macro P = 5/6, macro R = Acc = 5/6, macro F1 = 4/5, micro Acc 4/5; kappa: po 4/5, pe = (2*3 + 3*2)/25 = 12/25 -> (8/25)/(13/25) = 8/13
#   AUROC (class 1 scores p1 increasing in l1 - l0): positives {1, 2, -.5} vs negatives {-3, -1}: all 6 pairs won -> 1 (both classes)
C_expect = np.array([5 / 6, 1.0, 5 / 6, 5 / 6, 4 / 5, 8 / 13, 4 / 5])

CASES = [("A multiclass + absent class + tie", A_probs, A_labels, 3, False, A_expect),
         ("B binary task", B_logits, B_labels, 2, True, B_expect),
         ("C two-class softmax", C_logits, C_labels, 2, False, C_expect)]
