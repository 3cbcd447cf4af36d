function S = update_candidate(S, valid_key_candidates, image, K, hyper_paras) %#ok<INUSL>
% Drop-in for Continuous_operation/update_candidate.m: the O(N*M) redundancy test runs on the GPU.
R_W_C = reshape(S.est_rot(:, end), [3, 3]);
t_W_C = S.est_trans(:, end);
loc = valid_key_candidates.Location;
if ~isempty(S.C)
    keep = vohip_mex('candidate_filter', single(loc), S.C, S.P, hyper_paras.r_discard_redundant);
    new_valid = double(flipud(loc(keep, :)'));
else
    new_valid = double(flipud(loc'));
end
S.C = [S.C, new_valid];
S.F = [S.F, new_valid];
S.T = [S.T, repmat([R_W_C(:); t_W_C(:)], 1, size(loc, 1))];
end
