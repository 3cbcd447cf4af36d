function [T_W_C, init_points_valid, matched_points_valid] = pose_estimation_init(init_points, matched_points, K, hyper_paras)
% Drop-in for Initialization/pose_estimation_init.m ('fundamental' branch): the 2000-trial
% fundamental-matrix RANSAC and the pose disambiguation run on the GPU.
assert(strcmp(hyper_paras.sfm_pose, 'fundamental'), 'vohip: only sfm_pose = ''fundamental'' is built');
n = size(init_points, 1);
s = rng;                                 % the toolbox call would draw its trials from this stream
samples = zeros(8, 2000);
for t = 1:2000
    samples(:, t) = randperm(n, 8)';
end
rng(s); rand(1);                         % leave the stream advanced, as any RANSAC call would
[fRANSAC, inliers] = vohip_mex('fmatrix_ransac', double(init_points), double(matched_points), samples, 1e-2, 99.9);
essMat = K' * fRANSAC * K;
init_points_valid = init_points(inliers, :);
matched_points_valid = matched_points(inliers, :);
[R_W_C, t_W_C] = vohip_mex('relative_pose', essMat, K, double(init_points_valid), double(matched_points_valid));
T_W_C = [R_W_C, t_W_C'; zeros(1, 3), 1];
end
