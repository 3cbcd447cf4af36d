function [R_C_W, t_C_W, best_inlier_mask, max_num_inliers_history, num_iteration_history] ...
    = ransacLocalization(matched_query_keypoints, corresponding_landmarks, K)
% Drop-in for utils/exe7_Ransac/ransacLocalization.m (same signature, same outputs); the P3P
% hypotheses and their scoring run on the GPU through vohip_mex.
% RNG contract: the reference consumes one datasample call per executed iteration from the
% global stream (ransacLocalization.m:46-47).  Here a first chunk of 128 triples (the loop never
% executes fewer than 99 iterations, :129) is drawn with the same toolbox call; only when the
% adaptive rule asks for more (status 4) the chunk is doubled, continuing the same stream, and the
% call repeated with the longer list.  Afterwards the stream is rewound and exactly n_iter draws are
% replayed, so the global stream ends where the reference's would.
use_p3p = true;                       % the switch of ransacLocalization.m:9; false selects the DLT branch
if use_p3p
    k = 3; op = 'ransac_p3p';         % :19
else
    k = 6; op = 'ransac_dlt';         % :23
end
N = size(corresponding_landmarks, 2);
cap = 14999;                          % the loop cap of ransacLocalization.m:128 (15000, test '>')
s = rng;
samples = zeros(k, 0);
budget = 128;
while true
    first = size(samples, 2) + 1;
    samples(:, first:budget) = 0;
    for h = first:budget
        [~, samples(:, h)] = datasample(corresponding_landmarks, k, 2, 'Replace', false);
    end
    [R_C_W, t_C_W, mask, max_num_inliers_history, num_iteration_history, n_iter, status] = ...
        vohip_mex(op, double(matched_query_keypoints), double(corresponding_landmarks), K, samples);
    if status ~= 4 || budget >= cap
        break
    end
    budget = min(2 * budget, cap);
end
if n_iter ~= size(samples, 2)         % leave the stream after n_iter draws, not after the whole chunk
    rng(s);
    for h = 1:n_iter
        datasample(corresponding_landmarks, k, 2, 'Replace', false);
    end
end
if status == 0 || (status == 4 && any(mask))
    best_inlier_mask = mask;
else
    R_C_W = []; t_C_W = []; best_inlier_mask = zeros(1, N);
end
end
