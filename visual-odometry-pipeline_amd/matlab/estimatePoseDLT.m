function M_tilde = estimatePoseDLT(p, P, K)
% Drop-in for utils/exe7_Ransac/estimatePoseDLT.m (same signature): p [nx2] undistorted image points,
% P [nx3] 3-D points, K [3x3]; M_tilde [3x4] = [R_tilde | alpha*t], world -> camera.  The measurement
% matrix, its smallest right singular vector and the closest rotation are computed on the GPU.
M_tilde = vohip_mex('estimate_pose_dlt', double(p), double(P), double(K));
end
