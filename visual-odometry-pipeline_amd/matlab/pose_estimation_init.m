function [T_W_C, init_points_valid, matched_points_valid] = pose_estimation_init(init_points, matched_points, K, hyper_paras)
% Drop-in for Initialization/pose_estimation_init.m: the fundamental-matrix RANSAC and the pose
% disambiguation run on the GPU.  'fundamental' (:7-15): 2000 trials, Sampson distance 1e-2, confidence 99.9.
% 'essential' (:19): estimateEssentialMatrix is closed source (five-point solver in MSAC); it is defined here as the
% same RANSAC with that function's default settings (500 trials, Sampson distance 0.1 px^2, the call's confidence
% 99.9) followed by the nearest essential matrix to K' F K.
switch hyper_paras.sfm_pose
    case 'fundamental', trials = 2000; dist = 1e-2;
    case 'essential',   trials = 500;  dist = 0.1;
    otherwise, error('vohip:pose_estimation_init', 'sfm_pose %s is not built', hyper_paras.sfm_pose);
end
n = size(init_points, 1);
s = rng;                                 % the toolbox call would draw its trials from this stream
samples = zeros(8, trials);
for t = 1:trials
    samples(:, t) = randperm(n, 8)';
end
rng(s); rand(1);                         % leave the stream advanced, as any RANSAC call would
[fRANSAC, inliers] = vohip_mex('fmatrix_ransac', double(init_points), double(matched_points), samples, dist, 99.9);
essMat = K' * fRANSAC * K;
if strcmp(hyper_paras.sfm_pose, 'essential')
    [U, ~, V] = svd(essMat);
    essMat = U * diag([1 1 0]) * V';
end
init_points_valid = init_points(inliers, :);
matched_points_valid = matched_points(inliers, :);
[R_W_C, t_W_C] = vohip_mex('relative_pose', essMat, K, double(init_points_valid), double(matched_points_valid));
T_W_C = [R_W_C, t_W_C'; zeros(1, 3), 1];
end
