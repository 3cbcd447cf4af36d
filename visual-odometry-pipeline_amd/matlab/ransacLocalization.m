function [R_C_W, t_C_W, best_inlier_mask, max_num_inliers_history, num_iteration_history] ...
    = ransacLocalization(matched_query_keypoints, corresponding_landmarks, K)
% Drop-in for utils/exe7_Ransac/ransacLocalization.m (same signature, same outputs); the P3P
% hypotheses and their scoring run on the GPU through vohip_mex.
% RNG contract: the reference consumes one datasample call per executed iteration from the
% global stream.  Here the triples of the whole budget are drawn first with the same toolbox
% call, the stream is rewound, and exactly n_iter draws are replayed, so the global stream ends
% where the reference's would.
N = size(corresponding_landmarks, 2);
budget = 14999;                       % the loop cap of ransacLocalization.m:128 (15000, test '>')
s = rng;
samples = zeros(3, budget);
for h = 1:budget
    [~, samples(:, h)] = datasample(corresponding_landmarks, 3, 2, 'Replace', false);
end
[R_C_W, t_C_W, mask, max_num_inliers_history, num_iteration_history, n_iter, found] = ...
    vohip_mex('ransac_p3p', double(matched_query_keypoints), double(corresponding_landmarks), K, samples);
rng(s);
for h = 1:n_iter
    datasample(corresponding_landmarks, 3, 2, 'Replace', false);
end
if found
    best_inlier_mask = mask;
else
    R_C_W = []; t_C_W = []; best_inlier_mask = zeros(1, N);
end
end
