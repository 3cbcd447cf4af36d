function [features, valid_key_points] = genKeypoints(image, method, options)
% Drop-in for utils/genKeypoints.m for the natively built detectors ('Harris', 'MinEigen').
% features is [] (the continuous-operation loop discards it); valid_key_points is a cornerPoints
% object, so .Location / .Metric / indexing behave as before.
opts = struct('MinQuality', 0.01, 'FilterSize', 5);
for k = 1:2:numel(options)
    opts.(options{k}) = options{k + 1};
end
switch method
    case 'Harris',   m = 0;
    case 'MinEigen', m = 1;
    otherwise, error('vohip:genKeypoints', 'method %s is not built natively', method);
end
[loc, metric] = vohip_mex('detect_corners', image, opts.MinQuality, opts.FilterSize, 16, m);
features = [];
valid_key_points = cornerPoints(loc, 'Metric', metric);
end
