"""TEST INFRASTRUCTURE -- numpy restatement of the reference's VAR(2) identification (README.md:108-130), op for op:
    for i = PN+1:num_train: AA(i-PN, n*(j-1)+1 : n*j) = ad_acc(i-j, :) (j = 1..PN) ;  BB(i-PN, :) = ad_acc(i, :)
    PARA = (AA'*AA) \\ AA'*BB ;  A1 = PARA(1:n, :)' ;  A2 = PARA(n+1:2n, :)'
PARITY UNPINNED (the reference is MATLAB only and ships neither the data set nor outputs).  Checker only."""
import numpy as np


def identify_var2(ad_acc, num_train):
    ad = np.asarray(ad_acc, dtype=np.float64)
    n = ad.shape[1]
    PN = 2
    AA = np.zeros((num_train - PN, PN * n)); BB = np.zeros((num_train - PN, n))
    for i in range(PN, num_train):                 # MATLAB i = PN+1 .. num_train (1-based)
        for j in range(1, PN + 1):
            AA[i - PN, n * (j - 1):n * j] = ad[i - j]
        BB[i - PN] = ad[i]
    PARA = np.linalg.solve(AA.T @ AA, AA.T @ BB)
    return PARA[:n].T.copy(), PARA[n:2 * n].T.copy()
