function [S, B] = update_landmarks(S, B, KLT_tracker_C, image, K, hyper_paras)
% Drop-in for Continuous_operation/update_landmarks.m.  KLT_tracker_C may be a vision.PointTracker
% or a vohip.PointTracker; the epipolar filter, the angle test, the triangulation and the depth
% filter run on the GPU (one call each instead of one per candidate).
T_W_C = [reshape(S.est_rot(:, end), [3, 3]), S.est_trans(:, end); 0 0 0 1];
[pts, valid] = KLT_tracker_C(image);
matched = fliplr(pts(valid, :));
p1 = S.C(:, valid)';
% The toolbox call this replaces (update_landmarks.m:13-14) draws its trials from the global stream; how
% much of the stream it consumes is not documented, so the stream position after this function is a
% convention of this drop-in (500 randperm calls), not the toolbox's.
samples = zeros(8, 500);
for t = 1:500
    samples(:, t) = randperm(size(p1, 1), 8)';
end
[~, inl] = vohip_mex('fmatrix_ransac', double(p1), double(matched), samples, 5e-2, 99);
matched = matched(inl, :);
S.F = S.F(:, valid); S.F = S.F(:, inl);
S.T = S.T(:, valid); S.T = S.T(:, inl);
[status, X_new] = vohip_mex('candidate_triangulate', matched', S.F, S.T, T_W_C, K, ...
    hyper_paras.angle_threshold, hyper_paras.min_depth, hyper_paras.max_depth, hyper_paras.is_BA);
new = status == 1;
k = nnz(new);
cols = [X_new(:, new); B.new_idx + (0:k-1)];
S.X = [S.X, cols];
B.landmarks = [B.landmarks, cols];
B.new_idx = B.new_idx + k;
B.m = size(B.landmarks, 2);
S.P = [S.P, matched(new, :)'];
append = status ~= 0;
S.C = matched(~append, :)';
S.F = S.F(:, ~append);
S.T = S.T(:, ~append);
S.num_new = [S.num_new; nnz(append)];
end
